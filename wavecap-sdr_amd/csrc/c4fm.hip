// P25 C4FM demodulator bank -- placeholder until the kernel lands (this round).
#include "wh_common.h"
using namespace wh;
struct wh_c4fm_bank { int unused; };
extern "C" int wh_c4fm_bank_create(wh_c4fm_bank **, int, double, const float *, int, const float *, int,
                                   const float *, int) {
    return set_err(WH_E_ARG, "wh_c4fm_bank_create: not implemented in this build");
}
extern "C" int wh_c4fm_bank_run(wh_c4fm_bank *, const float *, size_t, size_t, uint8_t *, float *, size_t, int32_t *,
                                void *) {
    return set_err(WH_E_ARG, "wh_c4fm_bank_run: not implemented in this build");
}
extern "C" int wh_c4fm_bank_reset(wh_c4fm_bank *, void *) {
    return set_err(WH_E_ARG, "wh_c4fm_bank_reset: not implemented in this build");
}
extern "C" void wh_c4fm_bank_destroy(wh_c4fm_bank *) {}
