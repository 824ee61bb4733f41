// Shared by conv_igemm.hip and conv_igemm_splitk.hip: the implicit-GEMM tile (main loop + epilogue) and the descriptor
// check.  See conv_igemm.hip for the design notes.  (Two translation units: the split-K kernels are kept apart from the other
// twelve instantiations of the force-inlined tile to bound compile time and memory.)
#pragma once
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));


__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Epilogue of one 16-byte chunk (4 output channels of one output pixel m): v = scale*t + shift; [mask before add];
// v += add (+ add2); act; [mask after]; store through the output map.  One body, expanded in place by the tile epilogue
// and by the split-K finish kernel; it reads d, m, col, ncol, vec, t, sc, sh, y, add, mask, add2, HoWo from the scope it
// is expanded in.  (A macro so that both expansions are guaranteed in-line; a shared function would do as well.)
#define RN_EPI_CHUNK_BODY(GENERAL)                                                                              \
    do {                                                                                                         \
        float v[4] = {t.x * sc[0] + sh[0], t.y * sc[1] + sh[1], t.z * sc[2] + sh[2], t.w * sc[3] + sh[3]}; \
        int64_t off, aoff = -1, a2off = -1; \
        if (!GENERAL) { \
            off = m * d.Cout + col; \
            if (d.add_mode == 1) aoff = off; \
        } else { \
            const int n = (int)(m / HoWo); \
            const int rem = (int)(m - (int64_t)n * HoWo); \
            const int oh = rem / d.Wo, ow = rem - oh * d.Wo; \
            const int ph = oh * d.os + d.oo_h, pw = ow * d.os + d.oo_w; \
            const int64_t pix = (int64_t)ph * d.Wy + pw; \
            off = (int64_t)n * d.y_batch_stride + pix * d.Cout + col; \
            if (d.add_mode == 1) aoff = (int64_t)n * d.add_batch_stride + pix * d.Cout + col; \
            else if (d.add_mode == 2)                    /* nearest x2 upsample, cropped (D/model.py:88-108) */ \
                aoff = (int64_t)n * d.add_batch_stride + ((int64_t)(oh >> 1) * d.Wa + (ow >> 1)) * d.Cout + col; \
            if (d.add2_mode == 3 && ((ph | pw) & 1) == 0) \
                a2off = (int64_t)n * d.add2_batch_stride + ((int64_t)(ph >> 1) * d.Wa2 + (pw >> 1)) * d.Cout + col; \
        } \
        float mk[4] = {1.f, 1.f, 1.f, 1.f}, ad[4] = {0.f, 0.f, 0.f, 0.f}; \
        if (vec) { \
            if (d.mask_mode != 0) { const float4 q = *reinterpret_cast<const float4 *>(mask + off); mk[0] = q.x; mk[1] = q.y; mk[2] = q.z; mk[3] = q.w; } \
            if (aoff >= 0) { const float4 q = *reinterpret_cast<const float4 *>(add + aoff); ad[0] = q.x; ad[1] = q.y; ad[2] = q.z; ad[3] = q.w; } \
            if (a2off >= 0) { const float4 q = *reinterpret_cast<const float4 *>(add2 + a2off); ad[0] += q.x; ad[1] += q.y; ad[2] += q.z; ad[3] += q.w; } \
        } else { \
    _Pragma("unroll") \
            for (int j = 0; j < 4; ++j) { \
                if (j < ncol && d.mask_mode != 0) mk[j] = mask[off + j]; \
                if (j < ncol && aoff >= 0) ad[j] = add[aoff + j]; \
                if (j < ncol && a2off >= 0) ad[j] += add2[a2off + j]; \
            } \
        } \
    _Pragma("unroll") \
        for (int j = 0; j < 4; ++j) { \
            float u = v[j]; \
            if (d.mask_mode == 1) u = mk[j] > 0.f ? u : 0.f; \
            u += ad[j]; \
            if (d.act == 1) u = fmaxf(u, 0.f); \
            else if (d.act == 2) u = 1.0f / (1.0f + expf(-u)); \
            if (d.mask_mode == 2) u = mk[j] > 0.f ? u : 0.f; \
            v[j] = u; \
        } \
        if (vec) { \
            *reinterpret_cast<float4 *>(y + off) = make_float4(v[0], v[1], v[2], v[3]); \
        } else { \
    _Pragma("unroll") \
            for (int j = 0; j < 4; ++j) \
                if (j < ncol) y[off + j] = v[j]; \
        } \
    } while (0)

template <int WM, int WN, bool GENERAL, int BK>
__device__ __forceinline__ void conv_igemm_tile(const rn_conv_desc &d, const float *__restrict__ x,
                                                const float *__restrict__ w, float *__restrict__ y,
                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                const float *__restrict__ add, const float *__restrict__ mask,
                                                const float *__restrict__ add2, const int tile, const int ks_lo = 0,
                                                const int ks_hi = -1, float *__restrict__ partial = nullptr) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LDK = BK + 4;                            // padded LDS row, floats (conflict-free b128 reads)
    constexpr int CPK = BK / 4;                            // 16-byte chunks per staged row
    constexpr int RPS = 256 / CPK;                         // rows staged per pass
    constexpr int AR = BM / RPS, BR = BN / RPS;            // rows of A / B each thread stages per K-step
    static_assert(WM * WN == 4, "4 waves");
    __shared__ float lds[2][(BM + BN) * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (d.Cout + BN - 1) / BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int HoWo = d.Ho * d.Wo;
    const int64_t M = (int64_t)d.N * HoWo;
    const int K = d.kh * d.kw * d.Cin;
    const int Kpad = (K + 31) / 32 * 32;                   // packed weight rows are zero-padded to a multiple of 32
    const int nks = (ks_hi < 0 ? Kpad / BK : ks_hi) - ks_lo;   // K-steps of this workgroup: all of them, or one split-K slice
    const int dmask = (1 << d.div_shift) - 1;

    // ---- per-thread staging geometry: chunk column q (4 floats of K), rows srow + 32*i
    const int q = tid % CPK, srow = tid / CPK;
    const float *a_base[AR];
    int a_h[AR], a_w[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int64_t m = (int64_t)m0 + srow + RPS * i;
        if (m < M) {
            const int n = (int)(m / HoWo);
            const int rem = (int)(m - (int64_t)n * HoWo);
            const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
            a_base[i] = x + (int64_t)n * d.x_batch_stride;
            a_h[i] = oh * d.a + d.p;
            a_w[i] = ow * d.a + d.p_w;
        } else {
            a_base[i] = x;
            a_h[i] = -(1 << 28);                           // fails every bounds test
            a_w[i] = 0;
        }
    }
    const float *b_base[BR];
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + srow + RPS * i;
        b_ok[i] = n < d.Cout;
        b_base[i] = w + (int64_t)(b_ok[i] ? n : 0) * Kpad + 4 * q;
    }

    // two register sets: the loads of K-step ks+2 are issued while step ks is multiplied and step ks+1 waits in the
    // other set -- two K-steps (~4 us) of prefetch distance, enough for an HBM miss on inputs that fit neither L2 nor
    // the Infinity Cache (the 135x240 and larger levels)
    struct Stage { float4 ra[AR], rb[BR]; unsigned a_ok; };
    Stage sA, sB;
    // Fast path (Cin a multiple of the K-step, i.e. every layer but the 4-channel stem and channel-padded head
    // gradients): a K-step lies inside one filter tap, so the per-row input coordinate, bounds test and pixel offset
    // are recomputed only when the tap changes (every Cin/BK steps); in between a step is one add per row.
    const bool fast = (d.Cin % BK) == 0;
    int a_pix[AR];                                         // fast path: in-image offset of the row's pixel for the tap, <0 = zero
    int f_r = 0, f_s = 0, f_c = 0;                         // tap (r,s) and channel offset of the NEXT step to load
    bool f_new = true;                                     // the first step computes its tap even when it starts mid-tap
    if (fast && ks_lo > 0) {
        const int k0 = ks_lo * BK, tap0 = k0 / d.Cin;
        f_c = k0 - tap0 * d.Cin;
        f_r = tap0 / d.kw;
        f_s = tap0 - f_r * d.kw;
    }
    auto load_step = [&](int ks_raw, Stage &S) {
        // always executed (a step index past the end is clamped and its data never read): loads inside conditional
        // blocks make the compiler's s_waitcnt accounting fall back to vmcnt(0)
        const int ks = ks_lo + (ks_raw < nks ? ks_raw : nks - 1);
        float4 (&ra)[AR] = S.ra;
        float4 (&rb)[BR] = S.rb;
        unsigned &a_ok = S.a_ok;
        // weights first: their addresses need no arithmetic, and every load of the step is then in flight together
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = *reinterpret_cast<const float4 *>(b_base[i] + ks * BK);   // row clamped above
        if (fast) {
            if (f_c == 0 || f_new) {                       // new tap (wave-uniform)
                f_new = false;
                const bool tap_ok = f_r < d.kh;
                const int hoff = f_r * d.b, woff = f_s * d.b;
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const int nh = a_h[i] + hoff, nw = a_w[i] + woff;
                    const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                    const bool ok = tap_ok && ((nh | nw) >= 0) && (((nh | nw) & dmask) == 0) && ih < d.Hi && iw < d.Wi;
                    a_pix[i] = ok ? (ih * d.Wi + iw) * d.Cin : -1;
                }
            }
            const int c = f_c + 4 * q;
            a_ok = 0;
#pragma unroll
            for (int i = 0; i < AR; ++i) {                  // unconditional, from a clamped offset; zeroed in store_step
                a_ok |= a_pix[i] >= 0 ? (1u << i) : 0u;
                ra[i] = *reinterpret_cast<const float4 *>(a_base[i] + ((a_pix[i] >= 0 ? a_pix[i] : 0) + c));
            }
            f_c += BK;
            if (f_c >= d.Cin) { f_c = 0; if (++f_s == d.kw) { f_s = 0; ++f_r; } }
        } else {
            const int k = ks * BK + 4 * q;
            const int tap = k / d.Cin;
            const int c0 = k - tap * d.Cin;
            const int r = tap / d.kw, s = tap - r * d.kw;
            const bool tap_ok = r < d.kh;
            const int hoff = r * d.b, woff = s * d.b;
            a_ok = 0;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int nh = a_h[i] + hoff, nw = a_w[i] + woff;
                const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                const bool ok = tap_ok && ((nh | nw) >= 0) && (((nh | nw) & dmask) == 0) && ih < d.Hi && iw < d.Wi;
                // offsets inside one image fit 32 bits (x_batch_stride < 2^31 floats is checked by the launcher)
                a_ok |= ok ? (1u << i) : 0u;
                ra[i] = *reinterpret_cast<const float4 *>(a_base[i] + (ok ? (ih * d.Wi + iw) * d.Cin + c0 : 0));
            }
        }
    };
    // Nothing in load_step may depend on a loaded value: a select, a clamp or a branch right behind a global load makes
    // the compiler wait for it on the spot -- one memory round trip per load instead of one per K-step.  Padding /
    // out-of-range rows are zeroed and the input ReLU applied here, on the way into LDS, after the MFMAs.
    auto store_step = [&](int buf, Stage &S) {
        float *A = lds[buf], *B = lds[buf] + BM * LDK;
        float4 (&ra)[AR] = S.ra;
        float4 (&rb)[BR] = S.rb;
        const unsigned a_ok = S.a_ok;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            float4 v = (a_ok & (1u << i)) ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (d.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(A + (srow + RPS * i) * LDK + 4 * q) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<float4 *>(B + (srow + RPS * i) * LDK + 4 * q) = b_ok[i] ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_step(0, sA);
    store_step(0, sA);
    __syncthreads();
    load_step(1, sA);

    const int frag = (lane & 31) * LDK + (lane >> 5) * 4;   // [row = lane&31][k = 4*(lane>>5)]
    auto multiply = [&](int buf) {
        const float *A = lds[buf] + (wm * 64) * LDK + frag;
        const float *B = lds[buf] + BM * LDK + (wn * 64) * LDK + frag;
#pragma unroll
        for (int st = 0; st < BK / 8; ++st) {
            const float4 a0 = *reinterpret_cast<const float4 *>(A + st * 8);
            const float4 a1 = *reinterpret_cast<const float4 *>(A + 32 * LDK + st * 8);
            const float4 b0 = *reinterpret_cast<const float4 *>(B + st * 8);
            const float4 b1 = *reinterpret_cast<const float4 *>(B + 32 * LDK + st * 8);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm][j], bv[tn][j], acc[tm][tn], 0, 0, 0);
        }
    };
    if constexpr (BK >= 32) {
        for (int ks = 0; ks < nks; ks += 2) {
            load_step(ks + 2, sB);
            multiply(0);
            store_step(1, sA);                              // step ks+1 (past the end: clamped data into the idle buffer)
            __syncthreads();
            if (ks + 1 >= nks) break;
            load_step(ks + 3, sA);
            multiply(1);
            store_step(0, sB);
            __syncthreads();
        }
    } else {
        // K-step 16 (small K, three workgroups per CU): one register set -- a second one would not fit 168 VGPRs
        for (int ks = 0; ks < nks; ++ks) {
            multiply(ks & 1);
            store_step((ks & 1) ^ 1, sA);
            __syncthreads();
            load_step(ks + 2, sA);
        }
    }

    // ---- epilogue: v = scale[c]*acc + shift[c]; [mask before add]; v += add (+ add2); act; [mask after]
    // The accumulator tile goes through LDS (the staging buffers are free after the last barrier) so that global
    // memory sees 16-byte accesses, 32 consecutive lanes on one 512-byte row segment: out, add and mask all move as
    // float4.  Accumulator element e of lane l is row (e&3) + 8*(e>>2) + 4*(l>>5), column l&31 of its 32x32 tile.
    constexpr int LDT = BN + 4;
    constexpr int EP = (BM * LDT > 2 * (BM + BN) * LDK) ? 2 : 1;   // passes when the tile outgrows the staging LDS
    constexpr int RP = BM / EP;                                    // tile rows per pass
    static_assert(RP * LDT <= 2 * (BM + BN) * LDK && RP % 64 == 0, "output tile pass must fit the staging LDS");
    float *T = &lds[0][0];
    constexpr int CPR = BN / 4, RPP = 256 / CPR;             // 16-byte chunks per tile row, rows per pass of stores
    const int c4 = tid % CPR;
    const int col = n0 + 4 * c4;
    const bool col_ok = col < d.Cout;
    const bool vec = (d.Cout & 3) == 0;                      // then col+3 < Cout and every row offset is 16-byte aligned
    const int ncol = vec ? 4 : (d.Cout - col < 4 ? d.Cout - col : 4);
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (col_ok && j < ncol && scale != nullptr) sc[j] = scale[col + j];
        if (col_ok && j < ncol && shift != nullptr) sh[j] = shift[col + j];
    }
#pragma unroll
    for (int pass = 0; pass < EP; ++pass) {
        if (pass) __syncthreads();
        if ((wm * 64) / RP == pass) {
            const int rbase = wm * 64 - pass * RP;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        T[(rbase + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * LDT + wn * 64 + tn * 32 + (lane & 31)] =
                            acc[tm][tn][e];
        }
        __syncthreads();
        for (int r = tid / CPR; r < RP; r += RPP) {
            const int64_t m = (int64_t)m0 + pass * RP + r;
            if (m >= M || !col_ok) break;
            const float4 t = *reinterpret_cast<const float4 *>(T + r * LDT + 4 * c4);
            if (partial != nullptr) {                        // split-K: raw partial tile, the finish kernel does the rest
                float *pp = partial + m * d.Cout + col;
                if (vec) *reinterpret_cast<float4 *>(pp) = t;
                else { const float tt[4] = {t.x, t.y, t.z, t.w}; for (int j = 0; j < ncol; ++j) pp[j] = tt[j]; }
                continue;
            }
            RN_EPI_CHUNK_BODY(GENERAL);
        }
    }
}




static inline int check_desc(const rn_conv_desc *d) {
    if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return RN_EINVAL;
    if (d->Cin < 4 || (d->Cin & 3)) return RN_EINVAL;                     // 16-byte chunks must not straddle taps
    if ((int64_t)d->Hi * d->Wi * d->Cin > 0x7fffffffLL) return RN_EINVAL; // in-image offsets are 32-bit
    if (d->kh <= 0 || d->kw <= 0 || d->div_shift < 0 || d->div_shift > 2) return RN_EINVAL;
    if (d->add_mode < 0 || d->add_mode > 2 || d->act < 0 || d->act > 2) return RN_EINVAL;
    if (d->mask_mode < 0 || d->mask_mode > 2) return RN_EINVAL;
    if (d->os < 1 || d->oo_h < 0 || d->oo_w < 0 || (d->add2_mode != 0 && d->add2_mode != 3)) return RN_EINVAL;
    if ((d->Ho - 1) * d->os + d->oo_h >= d->Hy || (d->Wo - 1) * d->os + d->oo_w >= d->Wy) return RN_EINVAL;
    if (d->os != 1 && d->add_mode == 2) return RN_EINVAL;
    return RN_OK;
}

