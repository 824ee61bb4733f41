// Shared device helpers for libretinanet_mi355x (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/retinanet_mi355x.h"

#define RN_WAVE 64

#define RN_LAUNCH_CHECK()                         \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

static inline int rn_blocks(int64_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

// Sum over the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RN_WAVE);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RN_WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RN_WAVE);
    return v;
}

// Block-wide sum of up to 4 values per thread for a block of NW waves.  `red` needs NW*4 floats of LDS.
// Result valid in thread 0.
template <int NW>
__device__ __forceinline__ void block_sum4(float v[4], float *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[w * 4 + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = 0.f;
            for (int k = 0; k < NW; ++k) s += red[k * 4 + i];
            v[i] = s;
        }
    }
}
