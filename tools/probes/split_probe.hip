// Probe for the split-operand fp32 GEMM idea: x (fp32) = h + m + l with h, m, l bf16 (RNE of the running residual), and
// a*b ~= ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm on v_mfma_f32_32x32x16_bf16 (dropped terms <= 2^-23 |ab|, 2^-25 rms).
// Measures cycles per K=16 step of a 64x64 wave tile (4 fragments of 8 floats from LDS, split, 24 MFMAs) against the
// same loop with (a) 24 MFMAs and no split, (b) the 32 fp32 MFMAs 32x32x2 of the native kernel; and checks that the
// v_dot2c_f32_bf16 form of the residual is exact.
//   hipcc --offload-arch=gfx950 -O3 split_probe.hip -o split_probe && ./split_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    bf16x2 hp = {(__bf16)x0, (__bf16)x1};
    h = __builtin_bit_cast(unsigned, hp);
    float r0, r1;
    if (MODE == 1) {
        r0 = x0 - __builtin_bit_cast(float, h << 16);
        r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    } else {
        const bf16x2 e0 = {(__bf16)-1.0f, (__bf16)0.0f}, e1 = {(__bf16)0.0f, (__bf16)-1.0f};
        r0 = __builtin_amdgcn_fdot2_f32_bf16(hp, e0, x0, false);
        r1 = __builtin_amdgcn_fdot2_f32_bf16(hp, e1, x1, false);
    }
    bf16x2 mp = {(__bf16)r0, (__bf16)r1};
    m = __builtin_bit_cast(unsigned, mp);
    float s0, s1;
    if (MODE == 1) {
        s0 = r0 - __builtin_bit_cast(float, m << 16);
        s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    } else {
        const bf16x2 e0 = {(__bf16)-1.0f, (__bf16)0.0f}, e1 = {(__bf16)0.0f, (__bf16)-1.0f};
        s0 = __builtin_amdgcn_fdot2_f32_bf16(mp, e0, r0, false);
        s1 = __builtin_amdgcn_fdot2_f32_bf16(mp, e1, r1, false);
    }
    bf16x2 lp = {(__bf16)s0, (__bf16)s1};
    l = __builtin_bit_cast(unsigned, lp);
}
template <int MODE>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    u32x4 hh, mm, ll;
#pragma unroll
    for (int j = 0; j < 4; ++j) { unsigned a, b, c; split_pair<MODE>(x[2 * j], x[2 * j + 1], a, b, c); hh[j] = a; mm[j] = b; ll[j] = c; }
    h = __builtin_bit_cast(bf16x8, hh); m = __builtin_bit_cast(bf16x8, mm); l = __builtin_bit_cast(bf16x8, ll);
}

// MODE 0: bf16 MFMAs only (fragments converted once, outside); 1: split by shifts + subtract; 2: split by dot2c; 3: fp32 MFMA
template <int MODE>
__global__ __launch_bounds__(256) void loop_kernel(const float *src, float *out, int iters) {
    __shared__ float lds[2][4096];                    // 2 buffers of 256 rows x 16 floats
    for (int i = threadIdx.x; i < 8192; i += 256) (&lds[0][0])[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float *S = lds[it & 1];
        float a[2][8], b[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra = (wave & 1) * 64 + t * 32 + r, rb = 128 + (wave >> 1) * 64 + t * 32 + r;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const float4 va = *reinterpret_cast<const float4 *>(S + ra * 16 + 4 * ((2 * st + g) ^ ((ra >> 2) & 3)));
                const float4 vb = *reinterpret_cast<const float4 *>(S + rb * 16 + 4 * ((2 * st + g) ^ ((rb >> 2) & 3)));
                a[t][4 * st] = va.x; a[t][4 * st + 1] = va.y; a[t][4 * st + 2] = va.z; a[t][4 * st + 3] = va.w;
                b[t][4 * st] = vb.x; b[t][4 * st + 1] = vb.y; b[t][4 * st + 2] = vb.z; b[t][4 * st + 3] = vb.w;
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
        } else {
            bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (MODE == 0) {
                    u32x4 q, p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { q[j] = __builtin_bit_cast(unsigned, a[t][2 * j]); p[j] = __builtin_bit_cast(unsigned, b[t][2 * j]); }
                    ah[t] = am[t] = al[t] = __builtin_bit_cast(bf16x8, q);
                    bh[t] = bm[t] = bl[t] = __builtin_bit_cast(bf16x8, p);
                } else {
                    split8<MODE>(a[t], ah[t], am[t], al[t]);
                    split8<MODE>(b[t], bh[t], bm[t], bl[t]);
                }
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[tm], bm[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bm[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// One fragment (8 floats) -> (h, m, l), shift/mask form.
struct S8 { bf16x8 h, m, l; };
__device__ __forceinline__ S8 sp8(const float (&x)[8]) { S8 s; split8<1>(x, s.h, s.m, s.l); return s; }
#define MFMA6(ACC, A, B)                                                           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).l, (B).h, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).h, (B).l, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).m, (B).m, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).m, (B).h, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).h, (B).m, ACC, 0, 0, 0);     \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).h, (B).h, ACC, 0, 0, 0);
#define GROUPS(N, V) _Pragma("unroll") for (int q_ = 0; q_ < N; ++q_) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, V, 0); }

__device__ __forceinline__ void read_frags(const float *S, int wave, int r, int g, float (&a)[2][8], float (&b)[2][8]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = (wave & 1) * 64 + t * 32 + r, rb = 128 + (wave >> 1) * 64 + t * 32 + r;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const float4 va = *reinterpret_cast<const float4 *>(S + ra * 16 + 4 * ((2 * st + g) ^ ((ra >> 2) & 3)));
            const float4 vb = *reinterpret_cast<const float4 *>(S + rb * 16 + 4 * ((2 * st + g) ^ ((rb >> 2) & 3)));
            a[t][4 * st] = va.x; a[t][4 * st + 1] = va.y; a[t][4 * st + 2] = va.z; a[t][4 * st + 3] = va.w;
            b[t][4 * st] = vb.x; b[t][4 * st + 1] = vb.y; b[t][4 * st + 2] = vb.z; b[t][4 * st + 3] = vb.w;
        }
    }
}

// MODE 4: within one K-step: split a0, b0; MFMAs(0,0) with the split of b1 between them; MFMAs(0,1) with the split of a1;
//         then the other twelve bare.   MODE 5: the fragments of step k+1 are read and split between the MFMAs of step k.
template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void sched_kernel(const float *src, float *out, int iters) {
    __shared__ float lds[2][4096];
    for (int i = threadIdx.x; i < 8192; i += 256) (&lds[0][0])[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (MODE == 4) {
        for (int it = 0; it < iters; ++it) {
            float a[2][8], b[2][8];
            read_frags(lds[it & 1], wave, r, g, a, b);
            const S8 a0 = sp8(a[0]), b0 = sp8(b[0]);
            __builtin_amdgcn_sched_barrier(0);
            const S8 b1 = sp8(b[1]);
            MFMA6(acc[0][0], a0, b0)
            GROUPS(6, 6)
            __builtin_amdgcn_sched_barrier(0);
            const S8 a1 = sp8(a[1]);
            MFMA6(acc[0][1], a0, b1)
            GROUPS(6, 6)
            __builtin_amdgcn_sched_barrier(0);
            MFMA6(acc[1][0], a1, b0)
            MFMA6(acc[1][1], a1, b1)
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        float a[2][8], b[2][8];
        read_frags(lds[0], wave, r, g, a, b);
        S8 a0 = sp8(a[0]), a1 = sp8(a[1]), b0 = sp8(b[0]), b1 = sp8(b[1]);
        for (int it = 0; it < iters; ++it) {
            read_frags(lds[(it + 1) & 1], wave, r, g, a, b);
            __builtin_amdgcn_sched_barrier(0);
            const S8 na0 = sp8(a[0]), nb0 = sp8(b[0]), na1 = sp8(a[1]), nb1 = sp8(b[1]);
            MFMA6(acc[0][0], a0, b0)
            MFMA6(acc[0][1], a0, b1)
            MFMA6(acc[1][0], a1, b0)
            MFMA6(acc[1][1], a1, b1)
            GROUPS(24, 6)
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// ---- hand-staged form: the split of four values (two pairs) in three pieces of six vector instructions, one piece after each MFMA,
// pinned by sched_barrier(0) (the scheduler otherwise gathers the vector work in front of the MFMAs).
struct Frag4 { u32x4 h[4], m[4], l[4]; };              // fragments a0, b0, a1, b1: (h, m, l) as 4 x 2 bf16
struct Quad { float x[4], r[4]; unsigned h[2], m[2]; };
__device__ __forceinline__ void stage_a(Quad &q) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bf16x2 hp = {(__bf16)q.x[2 * p], (__bf16)q.x[2 * p + 1]};
        q.h[p] = __builtin_bit_cast(unsigned, hp);
        q.r[2 * p] = __builtin_bit_cast(float, q.h[p] << 16);
        q.r[2 * p + 1] = __builtin_bit_cast(float, q.h[p] & 0xffff0000u);
    }
}
__device__ __forceinline__ void stage_b(Quad &q) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        q.x[2 * p] -= q.r[2 * p];
        q.x[2 * p + 1] -= q.r[2 * p + 1];
        const bf16x2 mp = {(__bf16)q.x[2 * p], (__bf16)q.x[2 * p + 1]};
        q.m[p] = __builtin_bit_cast(unsigned, mp);
    }
    q.r[0] = __builtin_bit_cast(float, q.m[0] << 16);
    q.r[1] = __builtin_bit_cast(float, q.m[0] & 0xffff0000u);
}
__device__ __forceinline__ void stage_c(Quad &q, unsigned &l0, unsigned &l1) {
    q.r[2] = __builtin_bit_cast(float, q.m[1] << 16);
    q.r[3] = __builtin_bit_cast(float, q.m[1] & 0xffff0000u);
    const bf16x2 lp0 = {(__bf16)(q.x[0] - q.r[0]), (__bf16)(q.x[1] - q.r[1])};
    const bf16x2 lp1 = {(__bf16)(q.x[2] - q.r[2]), (__bf16)(q.x[3] - q.r[3])};
    l0 = __builtin_bit_cast(unsigned, lp0);
    l1 = __builtin_bit_cast(unsigned, lp1);
}
#define BF8(V) __builtin_bit_cast(bf16x8, V)
// MFMA number N (0..23) of a step on fragments CUR: block (N / 12, (N / 6) % 2), product N % 6
#define STEP_MFMA(N, CUR)                                                                                               \
    {                                                                                                                   \
        constexpr int tm_ = (N) / 12, tn_ = ((N) / 6) % 2, pr_ = (N) % 6;                                               \
        const u32x4 *A_ = pr_ == 0 ? CUR.l : ((pr_ == 2 || pr_ == 3) ? CUR.m : CUR.h);                                  \
        const u32x4 *B_ = pr_ == 1 ? CUR.l : ((pr_ == 2 || pr_ == 4) ? CUR.m : CUR.h);                                  \
        acc[tm_][tn_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF8(A_[2 * tm_]), BF8(B_[2 * tn_ + 1]), acc[tm_][tn_], 0, 0, 0); \
    }
// piece N (0..23) of the split of the raw fragments RAW[4][8] into NXT: quad N / 3 (fragment (N / 3) / 2, words 2 * ((N / 3) % 2) ..)
#define STEP_SPLIT(N, NXT, RAW)                                                                                         \
    {                                                                                                                   \
        constexpr int qd_ = (N) / 3, f_ = qd_ / 2, w_ = 2 * (qd_ % 2);                                                  \
        if ((N) % 3 == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) quad.x[i_] = RAW[f_][2 * w_ + i_]; stage_a(quad); } \
        else if ((N) % 3 == 1) stage_b(quad);                                                                           \
        else { unsigned l0_, l1_; stage_c(quad, l0_, l1_); NXT.h[f_][w_] = quad.h[0]; NXT.h[f_][w_ + 1] = quad.h[1];    \
               NXT.m[f_][w_] = quad.m[0]; NXT.m[f_][w_ + 1] = quad.m[1]; NXT.l[f_][w_] = l0_; NXT.l[f_][w_ + 1] = l1_; } \
    }
#define STEP_ONE(N, CUR, NXT, RAW) STEP_MFMA(N, CUR) STEP_SPLIT(N, NXT, RAW) __builtin_amdgcn_sched_barrier(0);
#define STEP_ALL(CUR, NXT, RAW)                                                                                         \
    STEP_ONE(0, CUR, NXT, RAW) STEP_ONE(1, CUR, NXT, RAW) STEP_ONE(2, CUR, NXT, RAW) STEP_ONE(3, CUR, NXT, RAW)         \
    STEP_ONE(4, CUR, NXT, RAW) STEP_ONE(5, CUR, NXT, RAW) STEP_ONE(6, CUR, NXT, RAW) STEP_ONE(7, CUR, NXT, RAW)         \
    STEP_ONE(8, CUR, NXT, RAW) STEP_ONE(9, CUR, NXT, RAW) STEP_ONE(10, CUR, NXT, RAW) STEP_ONE(11, CUR, NXT, RAW)       \
    STEP_ONE(12, CUR, NXT, RAW) STEP_ONE(13, CUR, NXT, RAW) STEP_ONE(14, CUR, NXT, RAW) STEP_ONE(15, CUR, NXT, RAW)     \
    STEP_ONE(16, CUR, NXT, RAW) STEP_ONE(17, CUR, NXT, RAW) STEP_ONE(18, CUR, NXT, RAW) STEP_ONE(19, CUR, NXT, RAW)     \
    STEP_ONE(20, CUR, NXT, RAW) STEP_ONE(21, CUR, NXT, RAW) STEP_ONE(22, CUR, NXT, RAW) STEP_ONE(23, CUR, NXT, RAW)

template <int OCC>
__global__ __launch_bounds__(256, OCC) void staged_kernel(const float *src, float *out, int iters) {
    __shared__ float lds[2][4096];
    for (int i = threadIdx.x; i < 8192; i += 256) (&lds[0][0])[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    Frag4 F0, F1;
    Quad quad;
    float raw[4][8];                                    // a0, b0, a1, b1
    auto read_raw = [&](const float *S) {
        float a[2][8], b[2][8];
        read_frags(S, wave, r, g, a, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) { raw[0][j] = a[0][j]; raw[1][j] = b[0][j]; raw[2][j] = a[1][j]; raw[3][j] = b[1][j]; }
    };
    read_raw(lds[0]);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        bf16x8 h, m, l;
        split8<1>(raw[f], h, m, l);
        F0.h[f] = __builtin_bit_cast(u32x4, h); F0.m[f] = __builtin_bit_cast(u32x4, m); F0.l[f] = __builtin_bit_cast(u32x4, l);
    }
    for (int it = 0; it < iters; it += 2) {
        read_raw(lds[1]);
        __builtin_amdgcn_sched_barrier(0);
        STEP_ALL(F0, F1, raw)
        read_raw(lds[0]);
        __builtin_amdgcn_sched_barrier(0);
        STEP_ALL(F1, F0, raw)
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// MODE 7 / 8: the same loop with v_mfma_f32_16x16x32_bf16 (twice as many, half the cycles each) -- 7: no split, 8: with the split.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void loop16_kernel(const float *src, float *out, int iters) {
    __shared__ float lds[2][4096];
    for (int i = threadIdx.x; i < 8192; i += 256) (&lds[0][0])[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, g = lane >> 5;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
        float a[2][8], b[2][8];
        read_frags(lds[it & 1], wave, r, g, a, b);
        bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (MODE == 7) {
                u32x4 q, p;
#pragma unroll
                for (int j = 0; j < 4; ++j) { q[j] = __builtin_bit_cast(unsigned, a[t][2 * j]); p[j] = __builtin_bit_cast(unsigned, b[t][2 * j]); }
                ah[t] = am[t] = al[t] = __builtin_bit_cast(bf16x8, q);
                bh[t] = bm[t] = bl[t] = __builtin_bit_cast(bf16x8, p);
            } else {
                split8<1>(a[t], ah[t], am[t], al[t]);
                split8<1>(b[t], bh[t], bm[t], bl[t]);
            }
        }
        // 48 MFMAs of 16x16x32 = the FLOPs of 24 of 32x32x16 (operand shapes are immaterial to the timing)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int tm = (q >> 1) & 1, tn = q & 1;
            f32x4 &c0 = acc[2 * q], &c1 = acc[2 * q + 1];
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], bh[tn], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bl[tn], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[tm], bm[tn], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[tm], bh[tn], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bm[tn], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh[tn], c1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// exactness: per element, h + m + l (as doubles) against x, both split forms; and the two forms against each other
__global__ void exact_kernel(const float *x, unsigned *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned h1, m1, l1, h2, m2, l2;
    split_pair<1>(x[2 * i], x[2 * i + 1], h1, m1, l1);
    split_pair<2>(x[2 * i], x[2 * i + 1], h2, m2, l2);
    out[6 * i + 0] = h1; out[6 * i + 1] = m1; out[6 * i + 2] = l1; out[6 * i + 3] = h2; out[6 * i + 4] = m2; out[6 * i + 5] = l2;
}

static double bf(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int n = 1 << 20;
    float *hx = (float *)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const int e = rand() % 40 - 20;
        hx[i] = ldexpf((float)rand() / RAND_MAX * 2.f - 1.f, e);
    }
    float *dx; unsigned *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, (size_t)n * 3 * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact_kernel, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
    unsigned *ho = (unsigned *)malloc((size_t)n * 3 * 4);
    hipMemcpy(ho, dout, (size_t)n * 3 * 4, hipMemcpyDeviceToHost);
    double worst1 = 0, worst2 = 0; long differ = 0;
    for (int i = 0; i < n / 2; ++i)
        for (int half = 0; half < 2; ++half) {
            const double x = hx[2 * i + half];
            double s1 = 0, s2 = 0;
            for (int k = 0; k < 3; ++k) {
                s1 += bf((unsigned short)(ho[6 * i + k] >> (16 * half)));
                s2 += bf((unsigned short)(ho[6 * i + 3 + k] >> (16 * half)));
            }
            if (x != 0) { worst1 = fmax(worst1, fabs(s1 - x) / fabs(x)); worst2 = fmax(worst2, fabs(s2 - x) / fabs(x)); }
            for (int k = 0; k < 3; ++k) differ += ho[6 * i + k] != ho[6 * i + 3 + k];
        }
    printf("split exactness over %d values: max |h+m+l - x|/|x|  shifts+sub %.3g   dot2c %.3g   words differing between the forms %ld\n",
           n, worst1, worst2, differ);

    float *dout2; hipMalloc(&dout2, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char *names[9] = {"24 bf16 MFMAs, no split", "split (shift/and + v_pk_add)", "split (v_dot2c)", "32 fp32 MFMAs 32x32x2 (native)",
                            "split, interleaved within the step", "split one step ahead, interleaved", "split one step ahead, hand-staged", "48 bf16 MFMAs 16x16x32, no split", "48 bf16 MFMAs 16x16x32, split"};
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu)
        for (int mode = 0; mode < 9; ++mode) {
            if (mode == 2) continue;
            const int grid = 256 * wgs_per_cu;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                if (mode == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                if (mode == 2) hipLaunchKernelGGL(loop_kernel<2>, dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                if (mode == 3) hipLaunchKernelGGL(loop_kernel<3>, dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                if (mode == 4) hipLaunchKernelGGL((sched_kernel<4, 3>), dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                if (mode == 5) hipLaunchKernelGGL((sched_kernel<5, 2>), dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                if (mode == 6) hipLaunchKernelGGL((staged_kernel<2>), dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                if (mode == 7) hipLaunchKernelGGL(loop16_kernel<7>, dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                if (mode == 8) hipLaunchKernelGGL(loop16_kernel<8>, dim3(grid), dim3(256), 0, 0, dx, dout2, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            // fp32-equivalent FLOPs: 2 * 64 * 64 * 16 per wave and iteration
            const double fl = 2.0 * 64 * 64 * 16 * 4 * grid * (double)iters;
            printf("%d workgroup(s)/CU  %-34s %8.3f ms  %7.1f ns per K16 step per wave-slot  %7.1f TF fp32-equivalent\n", wgs_per_cu, names[mode], ms,
                   ms * 1e6 / iters / wgs_per_cu, fl / ms / 1e9);
        }
    return 0;
}
