#include "cbl_common.h"
#include <string.h>

CBL_EXPORT const char* cbl_version(void) { return "cbl_amd 0.1 gfx950 (hipcc, -ffp-contract=off)"; }

CBL_EXPORT int cbl_device_arch_ok(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
