// Shared device helpers for libretinanet_mi355x (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/retinanet_mi355x.h"

#define RN_WAVE 64

// Diagnostic builds.  The kernels carry compile-time knock-outs, stamps and ablations (timing only, most give wrong results) that the
// measurements in DESIGN.md / profiles/ were made with: RN_KO, RN_SPLIT_ABL, RN_STAMP, RN_SGB, RN_PIN_MFMA, RN_SPLIT_NBUF (conv_igemm_tile.h,
// mfma_split.h), RN_WINO_ABL (conv_wino.hip), P8_ABL / P8_STAGGER (conv_bf16_p8.hip), Q8_ABL (conv_fp8_p8.hip), RN_AMAX_KO (mfma_split.h),
// RN_MF16H_OCC / RN_MF16_KO (conv_igemm_mf16.hip).  They are honoured ONLY in a build with -DRN_EXPERIMENT=1 (tools/build_variant.sh adds it and writes
// the library beside the product one); any other build drops them here, before the files that test them are read.
#ifndef RN_EXPERIMENT
#define RN_EXPERIMENT 0
#endif
#if !RN_EXPERIMENT
#undef RN_KO
#undef RN_SPLIT_ABL
#undef RN_STAMP
#undef RN_SGB
#undef RN_PIN_MFMA
#undef RN_SPLIT_NBUF
#undef RN_WINO_ABL
#undef P8_ABL
#undef P8_STAGGER
#undef Q8_ABL
#undef RN_AMAX_KO
#undef RN_MF16H_OCC
#undef RN_MF16_KO
#undef RN_MF16_PF
#undef RN_MF16_STG
#undef RN_WG_KO
#endif

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

// Bounding-box reduction of a wave on the DPP crossbar (no LDS round trips): x1,y1 -> min, x2,y2 -> max over the 64
// lanes.  One v_min/v_max_f32_dpp per step: row_shr 1/2/4/8 scan inside each 16-lane row, then row_bcast:15 into rows
// 1,3 and row_bcast:31 into rows 2,3; lane 63 holds the result, returned wave-uniform.  Lanes without a DPP source are
// not written and keep their own value (neutral for min / max).  The four chains are interleaved, so every dependent
// DPP read is >= 2 instructions behind the write it needs (the gfx9 VALU-write -> DPP-read hazard); the s_nop guard
// the block against its neighbours, which the compiler cannot see into.
#define RN_DPP4(ctrl)                                      \
    "v_min_f32_dpp %0, %0, %0 " ctrl "\n"                  \
    "v_min_f32_dpp %1, %1, %1 " ctrl "\n"                  \
    "v_max_f32_dpp %2, %2, %2 " ctrl "\n"                  \
    "v_max_f32_dpp %3, %3, %3 " ctrl "\n"
__device__ __forceinline__ void wave_bbox_dpp(float &x1, float &y1, float &x2, float &y2) {
    asm volatile("s_nop 1\n"
                 RN_DPP4("row_shr:1 row_mask:0xf bank_mask:0xf")
                 RN_DPP4("row_shr:2 row_mask:0xf bank_mask:0xf")
                 RN_DPP4("row_shr:4 row_mask:0xf bank_mask:0xf")
                 RN_DPP4("row_shr:8 row_mask:0xf bank_mask:0xf")
                 RN_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 RN_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n"
                 : "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2));
    x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1), 63));
    y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y1), 63));
    x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x2), 63));
    y2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y2), 63));
}
__device__ __forceinline__ float rn_readlane_f(float v, int lane_uniform) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
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

// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2): id -> a work index such that every XCD gets ONE
// contiguous range of the work, in order.  Neighbouring items (pixel tiles that share halo rows, windows that overlap) then
// meet in one L2 instead of being fetched by up to eight.  For speed only: nothing depends on the placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ------------------------------------------------------------------------------------------------
// Buffer addressing and direct-to-LDS loads (conv_wgrad.hip, conv_igemm_tile.h).
typedef int v4i32 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor (stride 0): base, num_records in bytes, the gfx9-family dword 3 for untyped 32-bit data.
// A lane whose byte offset (voffset + soffset) is not below num_records reads 0.0 -- per dword, no memory access
// (tools/probes/buffer_probe.hip); the kernels use that as their zero-fill.
__device__ __forceinline__ v4i32 make_rsrc(const void *base, unsigned bytes) {
    const uint64_t b = (uint64_t)base;
    v4i32 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// One direct-to-LDS buffer load: lane l's 16 bytes at (descriptor base + voff + soff) land at LDS byte address
// lds_dst + 16*l; out-of-range lanes write zeros (tools/probes/lds_dma_probe.hip).  Issued as asm so that the compiler's
// waitcnt bookkeeping does not see it -- seen, it is taken to alias every later ds_read and fenced with vmcnt(0) BEFORE
// the MFMA phase it is meant to overlap.  The kernels count it themselves: rn_wait_dma() in front of the barrier that
// publishes the buffer.  M0 (the LDS base of the instruction) is compiler-reserved: saved, set and restored inside
// the one statement.
__device__ __forceinline__ void dma16(v4i32 rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void rn_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... for all but the newest N vector-memory operations (they return in order)
template <int N>
__device__ __forceinline__ void rn_wait_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// ------------------------------------------------------------------------------------------------
// Sign bits of an fp32 tensor (round 4): bit (e & 31) of word (e >> 5) says element e is > 0 -- one bit per element at the element's
// own offset, so every addressing scheme of the fp32 tensor (batch strides, parity classes, slices) carries over with offset >> 5.
// The backward pass needs of a ReLU output only this bit (the mask of its gradient): 1/32 of the bytes of re-reading the activation.
// A lane finishes four consecutive elements (a 16-byte chunk, offset a multiple of 4); the eight chunks of a word sit in eight
// consecutive lanes (an aligned group of 8: the epilogues' lanes run along the channels), which OR their nibbles over the DPP crossbar;
// the lane whose chunk starts the word stores it.  Every lane of the group must be active (the callers' bounds are uniform per group).
// (rn_conv_desc.mask_mode | RN_MASK_BITS, include/retinanet_mi355x.h: `mask` points to such words)
__device__ __forceinline__ void rn_sign_store(unsigned *bits, int64_t off, float a, float b, float c, float d) {
    unsigned w = ((unsigned)(a > 0.f) | ((unsigned)(b > 0.f) << 1) | ((unsigned)(c > 0.f) << 2) | ((unsigned)(d > 0.f) << 3)) << ((unsigned)off & 28u);
    w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x141, 0xF, 0xF, true);    // row_half_mirror: lane i <-> 7 - i of its group of 8
    if (((unsigned)off & 31u) == 0) bits[off >> 5] = w;
}
// the mask of four consecutive elements as 1.0 / 0.0: from the fp32 tensor itself (x > 0 is tested by the caller) or from its sign bits
__device__ __forceinline__ float4 rn_mask_load4(const float *mask, int64_t off, bool bits) {
    if (bits) {
        const unsigned n = reinterpret_cast<const unsigned *>(mask)[off >> 5] >> ((unsigned)off & 28u);
        return make_float4((n & 1u) ? 1.f : 0.f, (n & 2u) ? 1.f : 0.f, (n & 4u) ? 1.f : 0.f, (n & 8u) ? 1.f : 0.f);
    }
    return *reinterpret_cast<const float4 *>(mask + off);
}
// The same for a lane that finishes EIGHT consecutive elements (the bf16 kernels' 16-byte chunk): a byte per lane, four lanes per word.
__device__ __forceinline__ void rn_sign_store8(unsigned *bits, int64_t off, const float (&v)[8]) {
    unsigned b = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) b |= (unsigned)(v[j] > 0.f) << j;
    unsigned w = b << ((unsigned)off & 24u);
    w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    if (((unsigned)off & 31u) == 0) bits[off >> 5] = w;
}
// sign bits of elements off .. off + n - 1 (n = 4 or 8, off a multiple of n) in the low bits
__device__ __forceinline__ unsigned rn_sign_bits(const void *bits, int64_t off, int n) {
    return (reinterpret_cast<const unsigned *>(bits)[off >> 5] >> ((unsigned)off & (32u - (unsigned)n))) & ((1u << n) - 1u);
}
