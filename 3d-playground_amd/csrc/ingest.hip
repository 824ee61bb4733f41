// Frame ingest: uint8 HWC video frames -> normalised fp32 detector input, on device.
//
// Replaces F.to_tensor + F.normalize of the reference's loaders (util_track/mp_loader.py:239-243,
// perform_3D_detection_on_video_sequences.py:51-58), which run per frame on the host and ship a 24.9 MB fp32 CHW
// tensor per 1080p camera over PCIe; here the 6.2 MB uint8 frame is what travels and the conversion is one HBM pass:
//   v = (float(u8) / 255 - mean[c]) / std[c]        three separate fp32 operations, as torchvision performs them
//   (this file is compiled with -ffp-contract=off; a true division, not a reciprocal multiply: bit-identical to the CPU)
// layout 0: NCHW [B,3,H,W]  -- exactly the tensor the reference hands to the model
// layout 1: NHWC4 [B,H,W,4] -- what the stem convolution consumes (4th channel 0), skipping rn_nchw_to_nhwc4
// swap_rb: source channel 2-c feeds output channel c (the cvtColor(BGR2RGB) of the second caller).
//
// Roofline: HBM.  Per pixel 3 B read, 12 B (NCHW) or 16 B (NHWC4) written.  A lane takes 4 consecutive pixels: three
// dword loads (12 B) and three float4 stores, one per plane (NCHW); for NHWC4 a lane takes 4 pixels 256 apart so
// that each store instruction of a wave writes 1 KiB contiguously.
#include <stdint.h>

#include "common.h"

struct IngestArgs {
    const uint8_t *src;
    float *dst;
    int64_t hw;              // pixels per image
    int B, swap_rb, layout;
    float mean[3], stdv[3];
};

__device__ __forceinline__ float ingest_one(unsigned u8, float mean, float stdv) {
    const float t = (float)u8 / 255.0f;          // to_tensor
    return (t - mean) / stdv;                    // normalize: sub_, div_
}

// hw % 4 == 0: every image starts dword-aligned in the byte stream and float4-aligned in every output plane.
__global__ __launch_bounds__(256) void ingest_kernel4(const IngestArgs a) {
    const int64_t groups = a.hw >> 2;                                   // 4-pixel groups per image
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (g >= groups) return;
    const uint32_t *s = reinterpret_cast<const uint32_t *>(a.src + ((int64_t)b * a.hw + 4 * g) * 3);
    const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];                    // bytes p0c0 p0c1 p0c2 p1c0 | p1c1 p1c2 p2c0 p2c1 | p2c2 p3c0 p3c1 p3c2
    unsigned px[4][3] = {{w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u},
                         {w0 >> 24, w1 & 255u, (w1 >> 8) & 255u},
                         {(w1 >> 16) & 255u, w1 >> 24, w2 & 255u},
                         {(w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24}};
    float v[4][3];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[p][c] = ingest_one(px[p][a.swap_rb ? 2 - c : c], a.mean[c], a.stdv[c]);
    if (a.layout == 1) {
        float4 *o = reinterpret_cast<float4 *>(a.dst) + (int64_t)b * a.hw + 4 * g;
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p] = make_float4(v[p][0], v[p][1], v[p][2], 0.f);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            reinterpret_cast<float4 *>(a.dst + ((int64_t)b * 3 + c) * a.hw)[g] = make_float4(v[0][c], v[1][c], v[2][c], v[3][c]);
    }
}

// NHWC4: the output is 16 B per pixel, so the store is already a full float4 per pixel; what matters is that the 64
// lanes of a store instruction write 1 KiB contiguously.  A lane therefore takes pixels tid, tid+256, tid+512, tid+768
// of its block's 1024 (byte loads: the reads are 16 % of the traffic and every 64-B line is shared by ~21 lanes).
__global__ __launch_bounds__(256) void ingest_nhwc4_kernel(const IngestArgs a) {
    const int64_t p0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int b = blockIdx.y;
    const uint8_t *s = a.src + (int64_t)b * a.hw * 3;
    float4 *o = reinterpret_cast<float4 *>(a.dst) + (int64_t)b * a.hw;
    unsigned u[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                       // all loads first
        const int64_t p = p0 + 256 * k;
        const int64_t q = p < a.hw ? p : a.hw - 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) u[k][c] = s[q * 3 + c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t p = p0 + 256 * k;
        if (p < a.hw)
            o[p] = make_float4(ingest_one(u[k][a.swap_rb ? 2 : 0], a.mean[0], a.stdv[0]), ingest_one(u[k][1], a.mean[1], a.stdv[1]),
                               ingest_one(u[k][a.swap_rb ? 0 : 2], a.mean[2], a.stdv[2]), 0.f);
    }
}

// any size: one pixel per lane, byte loads
__global__ __launch_bounds__(256) void ingest_kernel1(const IngestArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= a.hw) return;
    const uint8_t *s = a.src + ((int64_t)b * a.hw + p) * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = ingest_one(s[a.swap_rb ? 2 - c : c], a.mean[c], a.stdv[c]);
    if (a.layout == 1) {
        reinterpret_cast<float4 *>(a.dst)[(int64_t)b * a.hw + p] = make_float4(v[0], v[1], v[2], 0.f);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.dst[((int64_t)b * 3 + c) * a.hw + p] = v[c];
    }
}

extern "C" int rn_frame_ingest(const uint8_t *frames, int B, int H, int W, int swap_rb, float mean0, float mean1,
                               float mean2, float std0, float std1, float std2, int layout, float *out, void *stream) {
    if (!frames || !out || B <= 0 || H <= 0 || W <= 0 || B > 65535 || (layout != 0 && layout != 1)) return RN_EINVAL;
    IngestArgs a;
    a.src = frames; a.dst = out; a.hw = (int64_t)H * W; a.B = B; a.swap_rb = swap_rb ? 1 : 0; a.layout = layout;
    a.mean[0] = mean0; a.mean[1] = mean1; a.mean[2] = mean2;
    a.stdv[0] = std0; a.stdv[1] = std1; a.stdv[2] = std2;
    hipStream_t s = (hipStream_t)stream;
    const bool aligned = (a.hw & 3) == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0;
    if (layout == 1)
        hipLaunchKernelGGL(ingest_nhwc4_kernel, dim3(rn_blocks(a.hw, 1024), B), dim3(256), 0, s, a);
    else if (aligned)
        hipLaunchKernelGGL(ingest_kernel4, dim3(rn_blocks(a.hw >> 2, 256), B), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(ingest_kernel1, dim3(rn_blocks(a.hw, 256), B), dim3(256), 0, s, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
