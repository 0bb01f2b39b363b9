// Library-level entry points of libwavehip.so.
#include "wh_common.h"

extern "C" int wh_abi_version(void) { return 1; }

extern "C" const char *wh_last_error(void) { return wh::err_buf(); }

extern "C" int wh_device_info(int *cu_count, int *lds_bytes, char *name, size_t name_len) {
    int dev = 0;
    WH_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    WH_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (name && name_len) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return WH_OK;
}
