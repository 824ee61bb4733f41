// Anchor generation on device.  Replaces Anchors.forward / generate_anchors / shift
// (reference D/anchors.py:21-40, 42-73, 109-129), which rebuilds 389 205 boxes in numpy on the host and
// copies them to the GPU on every forward (38.5 ms on 8 CPU threads).
//
// Numerics: the reference adds the fp64 cell centre to the fp64 zero-centred base box and rounds ONCE to
// fp32.  An all-fp32 evaluation differs in 60 063 of 1 556 820 elements at 1080p, so the add is done in fp64
// here too (HBM-bound kernel: one fp64 add per element is free).  The 45 base boxes are computed on the host
// with the very same libm calls numpy makes and travel as a kernel argument.
//
// Roofline: HBM store of A*16 B (6.23 MB at 1080p); one float4 per lane, fully coalesced.
#include <math.h>

#include "common.h"

struct AnchorArgs {
    double base[RN_LEVELS][9][4];
    int gw[RN_LEVELS];
    int64_t first[RN_LEVELS + 1];   // first anchor index of each level, [RN_LEVELS] = total
};

extern "C" int64_t rn_anchor_count(int height, int width) {
    int64_t n = 0;
    for (int l = 3; l < 3 + RN_LEVELS; ++l) {
        const int s = 1 << l;
        n += 9LL * ((height + s - 1) / s) * ((width + s - 1) / s);   // D/anchors.py:25
    }
    return n;
}

extern "C" void rn_anchor_base_boxes(double out[RN_LEVELS * 9 * 4]) {
    const double ratios[3] = {0.5, 1.0, 2.0};                                        // D/anchors.py:17
    const double scales[3] = {1.0, pow(2.0, 1.0 / 3.0), pow(2.0, 2.0 / 3.0)};        // D/anchors.py:19
    for (int li = 0; li < RN_LEVELS; ++li) {
        const double size = (double)(1 << (li + 3 + 2));                             // D/anchors.py:15
        int i = 0;
        for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s, ++i) {
                const double side = size * scales[s];
                const double area = side * side;
                const double w = sqrt(area / ratios[r]);                             // D/anchors.py:64
                const double h = w * ratios[r];                                      // D/anchors.py:65
                double *o = out + (li * 9 + i) * 4;
                o[0] = 0.0 - w * 0.5;                                                // D/anchors.py:68-69
                o[1] = 0.0 - h * 0.5;
                o[2] = w - w * 0.5;
                o[3] = h - h * 0.5;
            }
    }
}

__global__ __launch_bounds__(256) void anchors_kernel(float4 *__restrict__ out, AnchorArgs args) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= args.first[RN_LEVELS]) return;
    int li = 0;
#pragma unroll
    for (int l = 1; l < RN_LEVELS; ++l) li += (i >= args.first[l]);
    const int64_t local = i - args.first[li];
    const int a = (int)(local % 9);
    const int64_t cell = local / 9;
    const int gw = args.gw[li];
    const int col = (int)(cell % gw), row = (int)(cell / gw);
    const double stride = (double)(1 << (li + 3));
    const double cx = ((double)col + 0.5) * stride;                                  // D/anchors.py:110
    const double cy = ((double)row + 0.5) * stride;                                  // D/anchors.py:111
    const double *b = args.base[li][a];
    out[i] = make_float4((float)(b[0] + cx), (float)(b[1] + cy), (float)(b[2] + cx), (float)(b[3] + cy));
}

extern "C" int rn_anchors_fwd(float *out, int height, int width, void *stream) {
    if (height <= 0 || width <= 0 || out == nullptr) return RN_EINVAL;
    AnchorArgs args;
    rn_anchor_base_boxes(&args.base[0][0][0]);
    int64_t n = 0;
    for (int li = 0; li < RN_LEVELS; ++li) {
        const int s = 1 << (li + 3);
        const int gh = (height + s - 1) / s;
        args.gw[li] = (width + s - 1) / s;
        args.first[li] = n;
        n += 9LL * gh * args.gw[li];
    }
    args.first[RN_LEVELS] = n;
    hipLaunchKernelGGL(anchors_kernel, dim3(rn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4 *>(out), args);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
