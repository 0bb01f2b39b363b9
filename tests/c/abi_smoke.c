#include <stdio.h>
#include <string.h>
#include "wavehip.h"
int main(void) {
    wh_pfb *p = NULL;
    int rc = wh_pfb_create(&p, 7, 9, NULL);
    printf("%d %d %s\n", wh_abi_version(), rc, wh_last_error());
    return (rc == WH_E_ARG && p == NULL && strstr(wh_last_error(), "wh_pfb_create")) ? 0 : 1;
}
