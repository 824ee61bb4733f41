// Library identity and device check for libretinanet_mi355x.
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

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

// fp32 product mode of the convolution kernels (include/retinanet_mi355x.h): -1 = not set yet -> RN_FP32_MFMA from the environment.
static std::atomic<int> g_fp32_mode{-1};
extern "C" int rn_get_fp32_mfma(void) {
    if (g_fp32_mode < 0) {
        const char *e = getenv("RN_FP32_MFMA");
        g_fp32_mode = (e && strcmp(e, "native") == 0) ? RN_FP32_NATIVE
                    : (e && strcmp(e, "split") == 0)  ? RN_FP32_SPLIT
                    : (e && strcmp(e, "split3") == 0) ? RN_FP32_SPLIT3 : RN_FP32_DEFAULT;
    }
    return g_fp32_mode;
}
extern "C" int rn_set_fp32_mfma(int mode) {
    if (mode != RN_FP32_NATIVE && mode != RN_FP32_SPLIT && mode != RN_FP32_SPLIT3) return RN_EINVAL;
    g_fp32_mode = mode;
    return RN_OK;
}

// Convolutions whose reduction length kh*kw*Cin is below this stay on the fp32 MFMA kernels in RN_FP32_SPLIT mode.  Round 2: 192 --
// the short reductions (1x1 from 64 / 128 channels) lost 5-25 % to the split kernels' lower residency (three workgroups per CU
// instead of four; profiles/r02_fp32_split_by_shape.txt).  Round 3: 64, i.e. every layer of the detector -- with the activation operand
// split once per workgroup (SPLIT 3) the short reductions gain too: the training step 96.1 -> 97.4 images/s
// (RN_FP32_SPLIT_MIN_K = 192 / 128 / 64 / 192 in one call: 96.0 / 96.2 / 97.4 / 96.1; profiles/r03_split_min_k.txt).
extern "C" int rn_fp32_split_min_k(void) {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("RN_FP32_SPLIT_MIN_K");
        v = e ? atoi(e) : 64;
    }
    return v;
}

// Run-time options (include/retinanet_mi355x.h: RN_OPT_*).  INT_MIN = not set yet -> the environment variable, read ONCE at the first use,
// else the default.  The launch paths read the cached value (round 4: the per-launch getenv calls of the split kernels' selectors --
// up to four per convolution launch, ~700 launches per step -- are gone); tests and A/B tools switch with rn_set_option.
static std::atomic<int> g_opt[RN_OPT_COUNT];
static std::once_flag g_opt_once;
static const struct { const char *env; int dflt, max; } g_opt_def[RN_OPT_COUNT] = {
    {"RN_SPLITK", 1, 1},        {"RN_DETERMINISTIC", 0, 1},  {"RN_MF16", 1, 1},         {"RN_MF16_MIN", 1, INT_MAX},
    {"RN_WGRAD_ONCE", 1, 1},    {"RN_BF16_P8", 1, 2},        {"RN_FP8_P8", 1, 2},
};
// The whole table is filled from the environment exactly once (std::call_once: two host threads making their first launches together
// see one initialisation); after that a change of the environment is not seen -- mid-process changes go through rn_set_option.
static void opt_init() {
    for (int i = 0; i < RN_OPT_COUNT; ++i) {
        const char *e = getenv(g_opt_def[i].env);
        int v = e ? atoi(e) : g_opt_def[i].dflt;
        if (g_opt_def[i].max == 1) v = v != 0;
        g_opt[i].store(v < 0 ? 0 : (v > g_opt_def[i].max ? g_opt_def[i].max : v), std::memory_order_relaxed);
    }
}
extern "C" int rn_get_option(int option) {
    if (option < 0 || option >= RN_OPT_COUNT) return -1;
    std::call_once(g_opt_once, opt_init);
    return g_opt[option].load(std::memory_order_relaxed);
}
extern "C" int rn_set_option(int option, int value) {
    if (option < 0 || option >= RN_OPT_COUNT || value < 0 || value > g_opt_def[option].max) return RN_EINVAL;
    std::call_once(g_opt_once, opt_init);
    g_opt[option].store(value, std::memory_order_relaxed);
    return RN_OK;
}
