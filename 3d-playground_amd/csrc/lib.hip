// Library identity and device check for libretinanet_mi355x.
#include <string.h>

#include "common.h"

extern "C" const char *rn_version(void) { return "retinanet_mi355x 0.1 (gfx950)"; }

extern "C" int rn_check_device(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? RN_OK : RN_EINVAL;
}
