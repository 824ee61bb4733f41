// Shared by conv_igemm.hip and conv_igemm_splitk.hip: the implicit-GEMM tile (main loop + epilogue) and the descriptor
// check.  See conv_igemm.hip for the design notes.  (Two translation units: the split-K kernels are kept apart from the other
// twelve instantiations of the force-inlined tile to bound compile time and memory.)
#pragma once
#include <stdlib.h>

#include "common.h"
#include "mfma_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));



// Epilogue of one 16-byte chunk (4 output channels of one output pixel m): v = scale*t + shift; [mask before add];
// v += add (+ add2); act; [mask after]; store through the output map.  Three stages, expanded in place by the tile
// epilogue and by the split-K finish kernel (macros so that both expansions are guaranteed in-line); they read d, m,
// col, ncol, vec, t, sc, sh, y, add, mask, add2, HoWo from the scope they are expanded in.
//   RN_EPI_ADDR   -> int64_t off (output / mask / same-geometry addend), aoff, a2off (-1 = none)
//   RN_EPI_LOAD   -> float mk[4], ad[4] from mask / add / add2
//   RN_EPI_FINISH -> arithmetic and the store; rn_conv_desc.y_amax (an exponent table per image, mfma_split.h): the lane's running
//                    maximum of |stored value| in `float rn_am` of the scope, noted once per lane (rn_amax_note) after the last chunk
//                    when the scope's `bool rn_span` is false (all rows of the tile in one image: the usual case), else per chunk
//                    in the table of the chunk's own image m / HoWo
#define RN_EPI_ADDR(GENERAL)                                                                                     \
        int64_t off, aoff = -1, a2off = -1; \
        if (!GENERAL) { \
            off = m * d.Cout + col; \
            if (d.add_mode == 1) aoff = off; \
        } else { \
            /* m < N*Ho*Wo < 2^31 (check_desc): 32-bit unsigned divisions -- a 64-bit one is ~100 VALU instructions, per row */ \
            const unsigned mu_ = (unsigned)m; \
            const int n = (int)(mu_ / (unsigned)HoWo); \
            const int rem = (int)(mu_ - (unsigned)n * (unsigned)HoWo); \
            const int oh = (int)((unsigned)rem / (unsigned)d.Wo), ow = rem - oh * d.Wo; \
            const int ph = oh * d.os + d.oo_h, pw = ow * d.os + d.oo_w; \
            const int64_t pix = (int64_t)ph * d.Wy + pw; \
            off = (int64_t)n * d.y_batch_stride + pix * d.Cout + col; \
            if (d.add_mode == 1) aoff = (int64_t)n * d.add_batch_stride + pix * d.Cout + col; \
            else if (d.add_mode == 2)                    /* nearest x2 upsample, cropped (D/model.py:88-108) */ \
                aoff = (int64_t)n * d.add_batch_stride + ((int64_t)(oh >> 1) * d.Wa + (ow >> 1)) * d.Cout + col; \
            if (d.add2_mode == 3 && ((ph | pw) & 1) == 0) \
                a2off = (int64_t)n * d.add2_batch_stride + ((int64_t)(ph >> 1) * d.Wa2 + (pw >> 1)) * d.Cout + col; \
        }

#define RN_EPI_LOAD()                                                                                            \
        float mk[4] = {1.f, 1.f, 1.f, 1.f}, ad[4] = {0.f, 0.f, 0.f, 0.f}; \
        if (vec) { \
            if (d.mask_mode != 0) { const float4 q = rn_mask_load4(mask, off, (d.mask_mode & RN_MASK_BITS) != 0); mk[0] = q.x; mk[1] = q.y; mk[2] = q.z; mk[3] = q.w; } \
            if (aoff >= 0) { const float4 q = *reinterpret_cast<const float4 *>(add + aoff); ad[0] = q.x; ad[1] = q.y; ad[2] = q.z; ad[3] = q.w; } \
            if (a2off >= 0) { const float4 q = *reinterpret_cast<const float4 *>(add2 + a2off); ad[0] += q.x; ad[1] += q.y; ad[2] += q.z; ad[3] += q.w; } \
        } else { \
    _Pragma("unroll") \
            for (int j = 0; j < 4; ++j) { \
                if (j < ncol && d.mask_mode != 0) mk[j] = mask[off + j]; \
                if (j < ncol && aoff >= 0) ad[j] = add[aoff + j]; \
                if (j < ncol && a2off >= 0) ad[j] += add2[a2off + j]; \
            } \
        }

#define RN_EPI_FINISH()                                                                                          \
        float v[4] = {t.x * sc[0] + sh[0], t.y * sc[1] + sh[1], t.z * sc[2] + sh[2], t.w * sc[3] + sh[3]}; \
    _Pragma("unroll") \
        for (int j = 0; j < 4; ++j) { \
            float u = v[j]; \
            if ((d.mask_mode & 3) == 1) u = mk[j] > 0.f ? u : 0.f; \
            u += ad[j]; \
            if (d.act == 1) u = fmaxf(u, 0.f); \
            else if (d.act == 2) u = 1.0f / (1.0f + expf(-u)); \
            if ((d.mask_mode & 3) == 2) u = mk[j] > 0.f ? u : 0.f; \
            v[j] = u; \
        } \
        if (vec) { \
            *reinterpret_cast<float4 *>(y + off) = make_float4(v[0], v[1], v[2], v[3]); \
            if (d.sign_out != nullptr) rn_sign_store(reinterpret_cast<unsigned *>(d.sign_out), off, v[0], v[1], v[2], v[3]); \
            if (d.y_amax != nullptr) { \
                const float q_ = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); \
                if (rn_span) rn_amax_note(d.y_amax, (int64_t)((unsigned)m / (unsigned)HoWo), q_); else rn_am = fmaxf(rn_am, q_); \
            } \
        } else { \
            float q_ = 0.f; \
    _Pragma("unroll") \
            for (int j = 0; j < 4; ++j) \
                if (j < ncol) { y[off + j] = v[j]; q_ = fmaxf(q_, fabsf(v[j])); } \
            if (d.y_amax != nullptr) { if (rn_span) rn_amax_note(d.y_amax, (int64_t)((unsigned)m / (unsigned)HoWo), q_); else rn_am = fmaxf(rn_am, q_); } \
        }

#define RN_EPI_CHUNK_BODY(GENERAL) \
    do {                           \
        RN_EPI_ADDR(GENERAL)       \
        RN_EPI_LOAD()              \
        RN_EPI_FINISH()            \
    } while (0)

// LDS image of a staged operand: [rows][BK floats], rows unpadded (a direct-to-LDS load fills 1 KiB linearly: 64/CPK whole
// rows per wave instruction).  To keep the fragment reads conflict-free the 16-byte chunks of a row are permuted:
// logical chunk c of row r sits at position c ^ swz(r).  The permutation is applied on the SOURCE address of the load
// (the lane that fills position p of row r fetches chunk p ^ swz(r)) and again on the read.
//   BK = 32 (128-byte rows): swz = (r & 7) ^ ((r >> 3) & 7);   BK = 16 (64-byte rows): swz = (r >> 2) & 3.
// Checked against the servicing groups of ds_read_b128 (4 x 16 lanes: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the
// same + 32): within a group the 16 lanes hit 16 distinct 16-byte bank groups.
// The compiler sinks most of a K-step's MFMAs below the s_waitcnt vmcnt(0) + s_barrier that end the step (they only use
// registers), so the wait for the NEXT step's loads came ~200 cycles after their issue instead of a whole MFMA phase later:
// every wave stalled on memory once per K-step.  A scheduling barrier pins the MFMA phase in front of the wait.
#ifndef RN_PIN_MFMA
#define RN_PIN_MFMA 1
#endif
// RN_SGB 1: sched_group_barrier pattern that spreads the SPLIT 3 form's vector work between its MFMAs.  Measured equal or slightly
// slower than the compiler's own order (profiles/r03_split_a_once_ab.txt: 181.8 against 186.1 TF on 3x3 256->256): the K-step's
// parts -- MFMAs, loads, vector + LDS work -- ADD UP on this chip whatever their order inside a wave (r03_split_knockout.txt).
#ifndef RN_SGB
#define RN_SGB 0
#endif
// Diagnostic build only (-DRN_STAMP=1, tools/stamp_split.sh): s_memtime stamps between the parts of a SPLIT 3 K-step, summed per
// wave in scalar registers and added into rn_stamps[] once after the loop.  The stamps' fences forbid overlaps the real kernel has:
// read the SHARES.  No product build executes a stamp.
#ifndef RN_STAMP
#define RN_STAMP 0
#endif
#if RN_STAMP
static __device__ unsigned long long rn_stamps[16];
#define RN_T(i)                                                                                                  \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t[i])::"memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
#else
#define RN_T(i)
#endif
#ifndef RN_KO
#define RN_KO 0
#endif
#if RN_PIN_MFMA
#define RN_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define RN_PIN() do {} while (0)
#endif

template <int BK>
__device__ __forceinline__ int lds_swz(int row) {
    return BK == 32 ? ((row & 7) ^ ((row >> 3) & 7)) : ((row >> 2) & 3);
}

// SPLIT != 0: the products go through the bf16 matrix cores as six split-operand MFMAs per 32x32x16 block (mfma_split.h)
// instead of eight v_mfma_f32_32x32x2_f32; activation staging, A fragment reads and epilogue are the same code.
//   SPLIT 1: w is the packed fp32 tensor, both operands are split in registers.
//   SPLIT 2: w is the PRE-SPLIT form of the packed tensor (rn_split_weights: [rows][Kpad/16][h,m,l][16] bf16, 96 bytes per
//            row and K-step), staged as three bf16 planes [BN rows][32 bytes] and read as ready MFMA operands: half the
//            vector ALU work of SPLIT 1.  Within a plane the two 16-byte chunks of a row swap places in rows 16-31 of
//            every 32 (position 2*row + (chunk ^ ((row >> 4) & 1))), which makes the ds_read_b128 of [row = lane & 31]
//            [chunk = lane >> 5] conflict-free (16 distinct 16-byte bank groups per servicing group of 16 lanes).
// TERMS 2 (SPLIT 2 only; rn_conv_desc.w_format 3; round 5): the fp16 two-term form of RN_FP32_SPLIT3 (mfma_split.h, second half) for the
// layers conv_igemm_mf16.hip does not take (at most 64 output channels, the 4-channel stem, channel counts that are no multiple of 32,
// the input-ReLU form): two B planes (64-byte records), the A fragments scaled by their rows' power of two (the image's amax table)
// inside the split, three v_mfma_f32_32x32x16_f16 per block, the inverse scales applied row by row / column by column in the epilogue.
// TERMS 1 (SPLIT 2 only; rn_conv_desc.w_format 2): products from the operands' FIRST bf16 terms only -- one MFMA instead of six: the
// arithmetic of the bf16 / fp8 engines (their fp32 stem), not of the fp32 path.
template <int WM, int WN, bool GENERAL, int BK, bool RELU = false, bool RAW = false, int SPLIT = 0, int TERMS = 3>
__device__ __forceinline__ void conv_igemm_tile(const rn_conv_desc &d, const float *__restrict__ x,
                                                const float *__restrict__ w, float *__restrict__ y,
                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                const float *__restrict__ add, const float *__restrict__ mask,
                                                const float *__restrict__ add2, const int tile, const int ks_lo = 0,
                                                const int ks_hi = -1, float *__restrict__ partial = nullptr) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int CPK = BK / 4;                            // 16-byte chunks per staged row
    constexpr int RPI = 64 / CPK;                          // rows one wave instruction fills
    constexpr int SUB = BK / 16;                           // split forms: 16-wide MFMA steps per staged K-step
    constexpr bool HALF = SPLIT == 2 && TERMS == 2;        // fp16 two-term products (RN_FP32_SPLIT3)
    constexpr int NP = HALF ? 2 : 3;                       // planes of the pre-split B tile
    constexpr int REC = 32 * NP, EB = 2 * NP;              // bytes of a pre-split record (16 values) / per weight element
    constexpr int BPL = BN * 8;                            // SPLIT 2 / 3: floats' worth of one 16-bit plane of the B tile (BN rows x 32 bytes)
    constexpr int APL = BM * 8;                            // SPLIT 3: the same for the A tile
    constexpr int NBI = SUB * NP * BN / 32;                // SPLIT 2 / 3: wave instructions that fill the B planes: (sub-step, plane, 32 rows)
    constexpr bool PB = SPLIT >= 2;                        // B arrives pre-split
    constexpr int IA = SPLIT == 3 ? 1 : BM / RPI / 4, IB = PB ? (NBI + 3) / 4 : BN / RPI / 4;    // instructions per wave per K-step and operand (SPLIT 3: A rows per THREAD)
    constexpr int BOFF = SPLIT == 3 ? 3 * APL : BM * BK;   // floats: where the B part of a buffer starts
    constexpr int STEP = SPLIT == 3 ? 3 * (APL + BPL) : (SPLIT == 2 ? BM * BK + SUB * NP * BPL : (BM + BN) * BK);   // floats per buffer: A rows (planes), then B rows (planes)
    constexpr int UPT = BM * 2 / 256;                      // SPLIT 3: (row, 8-value half) units of the A tile per thread
    static_assert(SPLIT != 3 || (BK == 16 && UPT == 1), "SPLIT 3: 16-wide K-steps, 128-row tiles");
    static_assert(TERMS == 3 || SPLIT == 2, "one- and two-term products: the pre-split-weights form only");
    constexpr int LDT = BN + 4;                            // epilogue: padded output tile row
#ifndef RN_SPLIT_NBUF
#define RN_SPLIT_NBUF 2
#endif
    // LDS ring: two buffers (see the K loop for why not three).  The split forms spend 2.7x less time in the MFMAs of a K-step,
    // so a third buffer was tried for them again (-DRN_SPLIT_NBUF=3): faster on the long-K shapes in isolation (3x3 256->256 at
    // P4 166 -> 189 TF), slower on the short ones (residency: 60 KB of LDS = two workgroups per CU), and per training step
    // 89.2 against 90.6 images/s on the same GPU.
    constexpr int NBUF = SPLIT ? RN_SPLIT_NBUF : 2;
    constexpr int EP = (BM * LDT > NBUF * STEP) ? 2 : 1;   // epilogue passes when the output tile outgrows the staging LDS
    constexpr int RP = BM / EP;                            // tile rows per pass
    constexpr int LDSF = NBUF * STEP > RP * LDT ? NBUF * STEP : RP * LDT;
    static_assert(WM * WN == 4 && RP % 64 == 0 && IA >= 1 && IB >= 1, "tile shape");
    __shared__ float lds[LDSF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ntn = (d.Cout + BN - 1) / BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int HoWo = d.Ho * d.Wo;
    const int64_t M = (int64_t)d.N * HoWo;
    const int K = d.kh * d.kw * d.Cin;
    const int Kpad = (K + 31) / 32 * 32;                   // packed weight rows are zero-padded to a multiple of 32
    const int nks = (ks_hi < 0 ? Kpad / BK : ks_hi) - ks_lo;   // K-steps of this workgroup: all of them, or one split-K slice
    const int dmask = (1 << d.div_shift) - 1;

    // ---- buffer descriptors (scalar).  Out-of-range offsets read as zero: that is the zero padding of the convolution,
    // the rows past M and the weight rows past Cout -- no clamps, no masks, no selects on the data.
    //   x: from the first image this tile touches (a tile spans few images; the launcher checks that their byte span
    //      stays below 2 GiB, so offsets are plain int32 and 0x80000000 is always out of range);   w: the whole tensor.
    const int n_first = (int)(m0 / HoWo);
    const int64_t x_floats = ((int64_t)d.N - 1 - n_first) * d.x_batch_stride + (int64_t)d.Hi * d.Wi * d.Cin;
    const v4i32 rs_a = make_rsrc(x + (int64_t)n_first * d.x_batch_stride,
                                 (unsigned)(x_floats * 4 > 0x7FFFFFFF ? 0x7FFFFFFF : x_floats * 4));
    // per-image weights (w_batch_stride, in elements): a batch of GEMMs.  The pre-split form has 6 bytes per element.
    const v4i32 rs_b = PB
        ? make_rsrc(reinterpret_cast<const char *>(w) + (int64_t)n_first * d.w_batch_stride * EB, (unsigned)((int64_t)d.Cout * Kpad * EB))
        : make_rsrc(w + (int64_t)n_first * d.w_batch_stride, (unsigned)((int64_t)d.Cout * Kpad * 4));

    // ---- per-lane staging geometry: instruction j of this wave fills rows (wave*I + j)*RPI .. +RPI-1 of the operand;
    // the lane fills position pos of row rsub of them, i.e. fetches logical chunk pos ^ swz(row)
    const int pos = lane % CPK, rsub = lane / CPK;
    int a_h[IA], a_w[IA], a_img[IA], a_c[IA];              // row's input origin, its image's byte offset, chunk's first channel
    unsigned a_voff[IA];                                   // fast path: byte offset for the current tap (0x80000000 = zero row)
    const int rel0 = m0 - n_first * HoWo;
#pragma unroll
    for (int j = 0; j < IA; ++j) {
        // SPLIT 3: thread t owns (row t >> 1, values 8 * (t & 1) .. + 7 of the K-step) of the A tile: one MFMA operand chunk
        const int row = SPLIT == 3 ? tid >> 1 : (wave * IA + j) * RPI + rsub;
        a_c[j] = SPLIT == 3 ? 8 * (tid & 1) : 4 * (pos ^ lds_swz<BK>(row));
        if ((int64_t)m0 + row < M) {
            const unsigned rel = (unsigned)(rel0 + row);
            const unsigned n = rel / (unsigned)HoWo;
            const unsigned rem = rel - n * (unsigned)HoWo;
            const unsigned oh = rem / (unsigned)d.Wo, ow = rem - oh * (unsigned)d.Wo;
            a_img[j] = (int)((int64_t)n * d.x_batch_stride * 4);
            a_h[j] = (int)oh * d.a + d.p;
            a_w[j] = (int)ow * d.a + d.p_w;
        } else {
            a_img[j] = 0;
            a_h[j] = -(1 << 28);                           // fails every bounds test
            a_w[j] = 0;
        }
        a_voff[j] = 0x80000000u;
    }
    unsigned b_voff[IB];                                   // fixed for the whole kernel; the K-step advance is scalar
#pragma unroll
    for (int j = 0; j < IB; ++j) {
        if constexpr (PB) {
            // instruction q fills 32 rows of one plane: lane -> (row, position), fetches chunk position ^ ((row >> 4) & 1)
            const int q = wave * IB + j, sub = q / (NP * BN / 32), plane = (q / (BN / 32)) % NP, row = (q % (BN / 32)) * 32 + (lane >> 1);
            const int n = n0 + row;
            b_voff[j] = (q < NBI && n < d.Cout) ? (unsigned)(n * Kpad * EB + sub * REC + plane * 32 + (((lane & 1) ^ ((row >> 4) & 1)) << 4)) : 0x80000000u;
        } else {
            const int row = (wave * IB + j) * RPI + rsub;
            const int n = n0 + row;
            b_voff[j] = n < d.Cout ? (unsigned)((n * Kpad + 4 * (pos ^ lds_swz<BK>(row))) * 4) : 0x80000000u;
        }
    }

    // Fast path (Cin a multiple of the K-step, i.e. every layer but the 4-channel stem and channel-padded head
    // gradients): a K-step lies inside one filter tap, so the per-row input coordinate, bounds test and byte offset
    // are recomputed only when the tap changes (every Cin/BK steps); in between a step costs the vector ALU nothing --
    // the channel advance rides in the scalar offset of the load.
    const bool fast = (d.Cin % BK) == 0;
    const bool stem = d.Cin == 4 && d.kw == 8;             // the 7x7 stem as the engine packs it (NHWC4 input, taps padded to 8)
    int f_r = 0, f_s = 0, f_c = 0;                         // tap (r,s) and channel offset of the NEXT step to load
    bool f_new = true;                                     // the first step computes its tap even when it starts mid-tap
    if (fast && ks_lo > 0) {
        const int k0 = ks_lo * BK, tap0 = k0 / d.Cin;
        f_c = k0 - tap0 * d.Cin;
        f_r = tap0 / d.kw;
        f_s = tap0 - f_r * d.kw;
    }
    const unsigned lds0 = lds_addr(lds);
    // Issue the loads of K-step ks (relative to ks_lo) into buffer buf.  The caller's barrier has freed that buffer.
    auto dma_step = [&](int ks_rel, int buf) {
        const int ks = ks_lo + ks_rel;
        const unsigned A = lds0 + (unsigned)((buf * STEP + (wave_u * IA) * RPI * BK) * 4);
        const unsigned B = lds0 + (unsigned)((buf * STEP + BM * BK + (wave_u * IB) * RPI * BK) * 4);
#pragma unroll
        for (int j = 0; j < IB; ++j) {
            if constexpr (PB) {
                if (4 * IB <= NBI || wave_u * IB + j < NBI) dma16(rs_b, lds0 + (unsigned)((buf * STEP + BOFF) * 4 + (wave_u * IB + j) * 1024), b_voff[j], (unsigned)(ks * SUB * REC));
            } else {
                dma16(rs_b, B + j * (RPI * BK * 4), b_voff[j], (unsigned)(ks * BK * 4));
            }
        }
        if constexpr (SPLIT == 3) return;                   // the A operand goes through registers: load_a / split_a below
        if (fast) {
            if (f_c == 0 || f_new) {                       // new tap (wave-uniform)
                f_new = false;
                const bool tap_ok = f_r < d.kh;
                const int hoff = f_r * d.b, woff = f_s * d.b;
#pragma unroll
                for (int j = 0; j < IA; ++j) {
                    const int nh = a_h[j] + hoff, nw = a_w[j] + woff;
                    const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                    const bool ok = tap_ok & ((nh | nw) >= 0) & (((nh | nw) & dmask) == 0) & (ih < d.Hi) & (iw < d.Wi);
                    a_voff[j] = ok ? (unsigned)(a_img[j] + ((ih * d.Wi + iw) * d.Cin + a_c[j]) * 4) : 0x80000000u;
                }
            }
#pragma unroll
            for (int j = 0; j < IA; ++j) dma16(rs_a, A + j * (RPI * BK * 4), a_voff[j], (unsigned)(f_c * 4));
            f_c += BK;
            if (f_c >= d.Cin) { f_c = 0; if (++f_s == d.kw) { f_s = 0; ++f_r; } }
        } else {
#pragma unroll
            for (int j = 0; j < IA; ++j) {
                const int k = ks * BK + a_c[j];
                int tap, c0, r, s_;
                if (stem) {                                // 4 channels, 8-wide (padded) taps: the two divisions are shifts
                    tap = k >> 2; c0 = k & 3; r = tap >> 3; s_ = tap & 7;
                } else {
                    tap = k / d.Cin; c0 = k - tap * d.Cin; r = tap / d.kw; s_ = tap - r * d.kw;
                }
                const int nh = a_h[j] + r * d.b, nw = a_w[j] + s_ * d.b;
                const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                const bool ok = (r < d.kh) & ((nh | nw) >= 0) & (((nh | nw) & dmask) == 0) & (ih < d.Hi) & (iw < d.Wi);
                dma16(rs_a, A + j * (RPI * BK * 4), ok ? (unsigned)(a_img[j] + ((ih * d.Wi + iw) * d.Cin + c0) * 4) : 0x80000000u, 0u);
            }
        }
    };

    // ---- SPLIT 3: the A operand is split ONCE per workgroup.  Each thread loads its 8 consecutive values of one row into
    // registers (two 16-byte buffer loads; the range check is the zero-fill, as for the direct-to-LDS loads), two K-steps
    // ahead of the MFMAs that consume them; one step ahead it splits them (mfma_split.h) and stores the three bf16 terms as
    // ready MFMA operand chunks into the buffer's A planes -- the image the pre-split weights have: plane [BM rows][32 bytes],
    // chunk position 2 * row + (half ^ ((row >> 4) & 1)).  The MFMA phase then only reads operands: half the vector ALU work
    // of splitting per wave (the two waves that share a row block each split it in the SPLIT 1 / 2 forms).  Fast path only
    // (Cin a multiple of 16: the launcher sends the other layers to SPLIT 2).
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v ar[2], arn[2];                                   // two register sets, used in turn (the K loop is unrolled by two)
    // (the compiler's buffer-load builtin wants its own resource type: same base, range and dword 3 as rs_a)
    const __amdgpu_buffer_rsrc_t rs_a3 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + (int64_t)n_first * d.x_batch_stride), (short)0,
        (int)(unsigned)(x_floats * 4 > 0x7FFFFFFF ? 0x7FFFFFFF : x_floats * 4), 0x00020000);
    // Branch-free, and without multiplications (the compiler wraps anything expensive behind a select into a branch, and the K
    // loop's body must stay ONE basic block for the vector work to be spread between the MFMAs): with div_shift == 0 (the
    // launcher's condition for this form, with kh * kw <= 24) the byte offset of tap (r, s) is a per-thread base plus a
    // scalar, and whether the tap falls inside the image is one bit of a per-thread mask.  Steps past the last tap find a
    // zero bit and load zeros.
    unsigned a_mask = 0;
    for (int r = 0, t = 0; r < d.kh; ++r)
        for (int s_ = 0; s_ < d.kw; ++s_, ++t) {
            const int ih = a_h[0] + r * d.b, iw = a_w[0] + s_ * d.b;
            a_mask |= (unsigned)(((ih | iw) >= 0) & (ih < d.Hi) & (iw < d.Wi)) << t;
        }
    const int a_base = a_img[0] + ((a_h[0] * d.Wi + a_w[0]) * d.Cin + a_c[0]) * 4;
    auto load_a = [&](f32x4v (&ar)[2]) {
        const int t = f_r * d.kw + f_s;                                         // scalar
        const int delta = (f_r * d.Wi + f_s) * d.b * d.Cin * 4;
        // valid = all ones if bit t of the mask is set (t <= 31: kh * kw <= 24 and at most two steps past the end), else zero;
        // plain bit arithmetic -- behind a select the compiler moves even this one addition into a branch
        const unsigned valid = 0u - ((a_mask >> (t & 31)) & 1u);
        const unsigned v = ((unsigned)(a_base + delta) & valid) | (0x80000000u & ~valid);
        ar[0] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_a3, (int)v, f_c * 4, 0));
        ar[1] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_a3, (int)(v + 16u), f_c * 4, 0));
        f_c += BK;
        const bool wrap = f_c >= d.Cin;                     // scalar selects, no branch
        f_c = wrap ? 0 : f_c;
        f_s += wrap ? 1 : 0;
        const bool wrap_s = f_s == d.kw;
        f_s = wrap_s ? 0 : f_s;
        f_r += wrap_s ? 1 : 0;
    };
    const int a_wr = 4 * (2 * (tid >> 1) + ((tid & 1) ^ ((tid >> 5) & 1)));   // floats: this thread's chunk within a plane ((row >> 4) & 1 = (tid >> 5) & 1)
    auto split_a = [&](int buf, const f32x4v (&ar)[2]) {
        float av[8] = {ar[0][0], ar[0][1], ar[0][2], ar[0][3], ar[1][0], ar[1][1], ar[1][2], ar[1][3]};
        if (RELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = fmaxf(av[j], 0.f);
        }
        const Split8 sp = split8(av);
        float *P = lds + buf * STEP + a_wr;
        *reinterpret_cast<bf16x8 *>(P) = sp.h;
        *reinterpret_cast<bf16x8 *>(P + APL) = sp.m;
        *reinterpret_cast<bf16x8 *>(P + 2 * APL) = sp.l;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment addresses (floats, within a buffer): row = lane & 31 of each 32-row MFMA tile, logical chunk
    // 2*st + (lane >> 5) of K-sub-step st, at its swizzled position
    int fa[2][BK / 8], fb[2][BK / 8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int st = 0; st < BK / 8; ++st) {
            const int ra = wm * 64 + t * 32 + (lane & 31), rb = wn * 64 + t * 32 + (lane & 31);
            // split forms: a lane's eight k are CONSECUTIVE (chunks 2g, 2g + 1: the order of a pre-split record's half)
            const int ch = SPLIT ? 4 * (st >> 1) + 2 * (lane >> 5) + (st & 1) : 2 * st + (lane >> 5);
            fa[t][st] = ra * BK + 4 * (ch ^ lds_swz<BK>(ra));
            fb[t][st] = BM * BK + rb * BK + 4 * (ch ^ lds_swz<BK>(rb));
        }
    int fbs[2], fas[2];                                     // SPLIT 2 / 3: plane 0 of this lane's B (SPLIT 3: and A) operand of tile t (floats)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rb = wn * 64 + t * 32 + (lane & 31), ra = wm * 64 + t * 32 + (lane & 31);
        fbs[t] = BOFF + 4 * (2 * rb + ((lane >> 5) ^ ((rb >> 4) & 1)));
        fas[t] = 4 * (2 * ra + ((lane >> 5) ^ ((ra >> 4) & 1)));
    }
    // TERMS 2: the power-of-two scale of each of this lane's two fragment rows (wm * 64 + t * 32 + (lane & 31)) from its image's amax
    // table, and the inverses of all BM rows in a table of their own for the epilogue (written by the waves with wn == 0)
    __shared__ float row_unscale[HALF ? BM : 1];
    float a_scale[2] = {1.f, 1.f};
    if constexpr (HALF) {
        const unsigned char *tb = reinterpret_cast<const unsigned char *>(d.x_amax);
        const void *tl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t m = (int64_t)m0 + wm * 64 + t * 32 + (lane & 31);
            tl[t] = m < M ? tb + (int64_t)((unsigned)m / (unsigned)HoWo) * d.x_amax_img_stride * RN_AMAX_BYTES : nullptr;
        }
        int e[2];
        if (d.x_amax_row_stride == 0) {                    // per image: its exponent table
            e[0] = rn_amax_exp_lanes(tl[0]);
            e[1] = __ballot(tl[1] != tl[0]) == 0ull ? e[0] : rn_amax_exp_lanes(tl[1]);
        } else {                                           // per row: a plain word (the Winograd stage: rows are tiles)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t m = (int64_t)m0 + wm * 64 + t * 32 + (lane & 31);
                e[t] = m < M ? (int)((reinterpret_cast<const unsigned *>(d.x_amax)[(int64_t)((unsigned)m % (unsigned)HoWo) * d.x_amax_row_stride] >> 23) & 0xffu) : 0;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int se = rn_f16_scale_exp_of(e[t]);
            a_scale[t] = rn_exp_to_float(se);
            if (wn == 0 && lane < 32) row_unscale[wm * 64 + t * 32 + lane] = rn_exp_to_float(254 - se);
        }
    }
    auto multiply = [&](int buf) {
        const float *S = lds + buf * STEP;
        if constexpr (HALF) {
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                SplitH8 sa[2], sb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 p0 = *reinterpret_cast<const float4 *>(S + fa[t][2 * sub]), p1 = *reinterpret_cast<const float4 *>(S + fa[t][2 * sub + 1]);
                    float av[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                    if (RELU) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) av[j] = fmaxf(av[j], 0.f);
                    }
                    sa[t] = split8h(av, a_scale[t]);
                    const float *Bp = S + fbs[t] + sub * NP * BPL;
                    sb[t].h = *reinterpret_cast<const f16x8 *>(Bp);
                    sb[t].l = *reinterpret_cast<const f16x8 *>(Bp + BPL);
                }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) RN_SPLITH_MFMA(acc[tm][tn], sa[tm], sb[tn]);
            }
            return;
        }
        if constexpr (SPLIT != 0) {
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                Split8 sa[2], sb[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if constexpr (SPLIT == 3) {
                        const float *Ap = S + fas[t];
                        sa[t].h = *reinterpret_cast<const bf16x8 *>(Ap);
                        sa[t].m = *reinterpret_cast<const bf16x8 *>(Ap + APL);
                        sa[t].l = *reinterpret_cast<const bf16x8 *>(Ap + 2 * APL);
                    } else {
                        const float4 p0 = *reinterpret_cast<const float4 *>(S + fa[t][2 * sub]), p1 = *reinterpret_cast<const float4 *>(S + fa[t][2 * sub + 1]);
                        float av[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                        if (RELU) {                             // input ReLU, on the fragments
#pragma unroll
                            for (int j = 0; j < 8; ++j) av[j] = fmaxf(av[j], 0.f);
                        }
                        sa[t] = split8(av);
                    }
                    if constexpr (PB) {
                        const float *Bp = S + fbs[t] + sub * NP * BPL;
                        sb[t].h = *reinterpret_cast<const bf16x8 *>(Bp);
                        sb[t].m = *reinterpret_cast<const bf16x8 *>(Bp + BPL);
                        sb[t].l = *reinterpret_cast<const bf16x8 *>(Bp + 2 * BPL);
                    } else {
                        const float4 q0 = *reinterpret_cast<const float4 *>(S + fb[t][2 * sub]), q1 = *reinterpret_cast<const float4 *>(S + fb[t][2 * sub + 1]);
                        const float bv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                        sb[t] = split8(bv);
                    }
                }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        if constexpr (TERMS == 1) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[tm].h, sb[tn].h, acc[tm][tn], 0, 0, 0);
                        else RN_SPLIT_MFMA(acc[tm][tn], sa[tm], sb[tn]);
                    }
            }
            return;
        }
#pragma unroll
        for (int st = 0; st < BK / 8; ++st) {
            float4 a0 = *reinterpret_cast<const float4 *>(S + fa[0][st]);
            float4 a1 = *reinterpret_cast<const float4 *>(S + fa[1][st]);
            const float4 b0 = *reinterpret_cast<const float4 *>(S + fb[0][st]);
            const float4 b1 = *reinterpret_cast<const float4 *>(S + fb[1][st]);
            if (RELU) {                                     // input ReLU, on the fragments (inside the MFMA shadow)
                a0.x = fmaxf(a0.x, 0.f); a0.y = fmaxf(a0.y, 0.f); a0.z = fmaxf(a0.z, 0.f); a0.w = fmaxf(a0.w, 0.f);
                a1.x = fmaxf(a1.x, 0.f); a1.y = fmaxf(a1.y, 0.f); a1.z = fmaxf(a1.z, 0.f); a1.w = fmaxf(a1.w, 0.f);
            }
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

    // ---- K loop over a ring of NBUF LDS buffers: the loads of step ks + NBUF-1 are issued before the MFMAs of step ks,
    // one barrier per K-step.  With three buffers the wait in front of the barrier is COUNTED (everything but the newest
    // step's IA + IB loads must have landed) and loads stay in flight across it.
    // The loop is rolled: the buffer offset is a scalar added to the eight fragment addresses inside the MFMA shadow.
    // Unrolled (immediates for the buffer addresses) the kernel took 168 registers, rolled it takes 108-118, and that
    // decides the configuration: two buffers = 34 KB of LDS and <= 128 registers = FOUR workgroups per CU.  Measured per
    // training step (igemm kernels): unrolled, 2 buffers, 3 per CU 118.6 ms; rolled, 3 buffers, 3 per CU 119.9; rolled,
    // 2 buffers, 4 per CU 112.0 -- residency beats prefetch distance, most of all on the small-K layers.
    constexpr int NLD = IA + IB;                           // loads one wave issues per K-step
    // (SPLIT 2 with a 64-wide B tile: the last wave has no B plane to fill, its newest step is IA loads)
    const bool b_loader = !PB || (wave_u + 1) * IB <= NBI;
    static_assert(!PB || NBI % IB == 0, "a wave fills IB plane blocks or none");
    auto wait_but_newest = [&](bool newest_in_flight) {
        if (NBUF > 2 && newest_in_flight) {
            if (b_loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IA) : "memory");
        } else rn_wait_dma();
    };
    if constexpr (SPLIT == 3) {
        static_assert(SPLIT != 3 || NBUF == 2, "SPLIT 3: two buffers");
#if RN_STAMP
        unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long st_c0, st_r0, st_c1, st_r1;       // shader clock / 100 MHz reference around the K loop: the clock the loop ran at
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c0), "=s"(st_r0)::"memory");
#endif
        load_a(ar); dma_step(0, 0); split_a(0, ar);         // (the compiler waits for the registers it loaded)
        load_a(ar);                                         // step 1 stays in registers until iteration 0 splits it
        rn_wait_dma();
        __syncthreads();
        // One K-step: `cur` holds the A values of step ks + 1 (loaded one iteration ago), `nxt` receives step ks + 2.
        // Unconditional (one basic block): past the last step the loads return zeros / stale weights into a buffer nobody reads,
        // and every one of them has landed (vmcnt(0) below) before the epilogue reuses the LDS.
        auto k_step = [&](int ks, int rb, f32x4v (&cur)[2], f32x4v (&nxt)[2]) {
            // the compiler's wait for `cur` HERE, where it is free (the step ended with vmcnt(0)): placed after the loads below
            // it would be a vmcnt(0) that also covers them -- the compiler does not see the direct-to-LDS loads
            asm volatile("" : "+v"(cur[0]), "+v"(cur[1]));
#if RN_STAMP
            unsigned long long st_t[7];
#endif
            RN_T(0);
#if !(RN_KO & 1)                                            // knock-outs (timing only, wrong results): 1 no B loads, 2 no A loads, 4 one MFMA of six
            dma_step(ks + 1, rb ^ 1);                       // B planes of step ks + 1; buffer rb ^ 1 was released by the last barrier
#endif
#if !(RN_KO & 2)
            load_a(nxt);                                    // A values of step ks + 2: a whole MFMA phase to arrive
#endif
            RN_T(1);
            // This step's operands FIRST: the compiler cannot tell the two buffers apart, so every LDS read that follows the
            // plane stores in program order waits for them -- and they wait for the whole split.
            Split8 sa[2], sb[2];
            {
                const float *S = lds + rb * STEP;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float *Ap = S + fas[t], *Bp = S + fbs[t];
                    sa[t].h = *reinterpret_cast<const bf16x8 *>(Ap);
                    sa[t].m = *reinterpret_cast<const bf16x8 *>(Ap + APL);
                    sa[t].l = *reinterpret_cast<const bf16x8 *>(Ap + 2 * APL);
                    sb[t].h = *reinterpret_cast<const bf16x8 *>(Bp);
                    sb[t].m = *reinterpret_cast<const bf16x8 *>(Bp + BPL);
                    sb[t].l = *reinterpret_cast<const bf16x8 *>(Bp + 2 * BPL);
                }
            }
            RN_T(2);
#if RN_KO & 32                                              // LDS-traffic experiment (timing only): 8 more operand reads per K-step
            u32x4 xd0, xd1;
            {
                const unsigned la = lds_addr(lds + rb * STEP + fas[0]);
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096\n\tds_read_b128 %0, %2 offset:8192\n\tds_read_b128 %1, %2 offset:12288\n\t"
                             "ds_read_b128 %0, %2 offset:16384\n\tds_read_b128 %1, %2 offset:20480\n\tds_read_b128 %0, %2 offset:2048\n\tds_read_b128 %1, %2 offset:6144"
                             : "=&v"(xd0), "=&v"(xd1) : "v"(la) : "memory");
            }
#endif
            split_a(rb ^ 1, cur);
            RN_T(3);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) RN_SPLIT_MFMA(acc[tm][tn], sa[tm], sb[tn]);
            RN_T(4);
#if RN_KO & 32
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xd0), "+v"(xd1)::"memory");
#endif
#if RN_SGB
            // Spread the split's vector work and the plane stores between the 24 MFMAs (32 cycles each, of which 24 are free
            // issue slots): left alone the compiler puts all of it in front of the first MFMA.
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);            // A registers of step ks + 2
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);           // this step's operands
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);            // A planes of step ks + 1
            __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
#endif
            RN_PIN();
            rn_wait_dma();                                  // B planes of step ks + 1 landed, A registers of step ks + 2 arrived
            RN_T(5);
            __syncthreads();                                // (and the A planes' ds_writes: the barrier waits for lgkmcnt)
            RN_T(6);
#if RN_STAMP
#pragma unroll
            for (int i = 0; i < 6; ++i) st_sum[i] += st_t[i + 1] - st_t[i];
#endif
        };
        for (int ks = 0; ks < nks; ks += 2) {
            k_step(ks, 0, ar, arn);
            if (ks + 1 < nks) k_step(ks + 1, 1, arn, ar);
        }
#if RN_STAMP
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c1), "=s"(st_r1)::"memory");
        if (lane == 0) {
            atomicAdd(&rn_stamps[10], st_c1 - st_c0);
            atomicAdd(&rn_stamps[11], st_r1 - st_r0);
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(&rn_stamps[i], st_sum[i]);
            atomicAdd(&rn_stamps[8], (unsigned long long)nks);
            atomicAdd(&rn_stamps[9], 1ull);
        }
#endif
    }
    if (SPLIT != 3 && nks > 0) dma_step(0, 0);
    if (SPLIT != 3 && NBUF > 2 && nks > 1) dma_step(1, 1);
    if (SPLIT != 3) { wait_but_newest(nks > 1); __syncthreads(); }
    int rb = 0, wb = NBUF - 1;                              // buffer read by this step / filled for step ks + NBUF-1
    for (int ks = 0; SPLIT != 3 && ks < nks; ++ks) {
        const bool more = ks + (NBUF - 1) < nks;
        if (more) dma_step(ks + (NBUF - 1), wb);
        multiply(rb);
        RN_PIN();
        wait_but_newest(more);                              // the next step has landed (this wave's part) ...
        __syncthreads();                                    // ... and everybody's; buffer rb is free
        rb = rb == NBUF - 1 ? 0 : rb + 1;
        wb = wb == NBUF - 1 ? 0 : wb + 1;
    }

    // ---- RAW: a plain GEMM (no scale / shift / addend / mask / activation, dense output: the Winograd GEMMs) stores
    // straight from the accumulators -- every instruction writes two full 128-byte row segments -- and skips the LDS
    // staging with its two barriers, which is a measurable share of a 16-step K loop.
    if constexpr (RAW) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int col = n0 + wn * 64 + tn * 32 + (lane & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t m = (int64_t)m0 + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (m < M && col < d.Cout) {
                        if constexpr (HALF) {                       // (row_unscale: written before the K loop's barriers)
                            y[m * d.Cout + col] = acc[tm][tn][e] * row_unscale[m - m0] *
                                                  d.w_unscale[(d.w_batch_stride != 0 ? (int64_t)n_first * d.Cout : 0) + col];
                        } else y[m * d.Cout + col] = acc[tm][tn][e];
                    }
                }
            }
        return;
    }

    // ---- epilogue: v = scale[c]*acc + shift[c]; [mask before add]; v += add (+ add2); act; [mask after]
    // The accumulator tile goes through LDS (the staging buffers are free after the last barrier) so that global
    // memory sees 16-byte accesses, 32 consecutive lanes on one 512-byte row segment: out, add and mask all move as
    // float4.  Accumulator element e of lane l is row (e&3) + 8*(e>>2) + 4*(l>>5), column l&31 of its 32x32 tile.
    float *T = lds;
    float rn_am = 0.f;                                       // largest |y| this lane stored (rn_conv_desc.y_amax)
    const int64_t m_last = (int64_t)m0 + BM - 1 < M ? (int64_t)m0 + BM - 1 : M - 1;
    const bool rn_span = (int)(m0 / HoWo) != (int)(m_last / HoWo);           // the tile's rows lie in more than one image (scalar)
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
        if (HALF && col_ok && j < ncol) sc[j] *= d.w_unscale[(d.w_batch_stride != 0 ? (int64_t)n_first * d.Cout : 0) + col + j];
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
        if (vec && partial == nullptr) {
            // The residual / mask operands of a GROUP of rows are requested first, unconditionally (rows past M re-read row
            // M-1), then the group is finished: one memory round trip per group instead of one per row.  With the loads
            // inside the row loop a workgroup paid ~16 serialised HBM round trips here -- on the small-K layers (four K-steps
            // per tile) more than its whole K loop.
            constexpr int NIT = RP / RPP, G = NIT % 4 == 0 ? 4 : (NIT % 2 == 0 ? 2 : 1);
            if (col_ok) {
#pragma unroll 1
                for (int g = 0; g < NIT; g += G) {
                    int64_t off_[G];
                    float4 mk_[G], ad_[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int64_t mr = (int64_t)m0 + pass * RP + tid / CPR + (g + i) * RPP;
                        const int64_t m = mr < M ? mr : M - 1;
                        RN_EPI_ADDR(GENERAL)
                        off_[i] = off;
                        mk_[i] = make_float4(1.f, 1.f, 1.f, 1.f);
                        ad_[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (d.mask_mode != 0) mk_[i] = rn_mask_load4(mask, off, (d.mask_mode & RN_MASK_BITS) != 0);
                        if (d.add_mode != 0) ad_[i] = *reinterpret_cast<const float4 *>(add + aoff);
                        (void)a2off;
                    }
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int r = tid / CPR + (g + i) * RPP;
                        const int64_t m = (int64_t)m0 + pass * RP + r;
                        if (m < M) {
                            float4 t = *reinterpret_cast<const float4 *>(T + r * LDT + 4 * c4);
                            if constexpr (HALF) { const float ru = row_unscale[pass * RP + r]; t.x *= ru; t.y *= ru; t.z *= ru; t.w *= ru; }
                            const int64_t off = off_[i];
                            float mk[4] = {mk_[i].x, mk_[i].y, mk_[i].z, mk_[i].w}, ad[4] = {ad_[i].x, ad_[i].y, ad_[i].z, ad_[i].w};
                            if (d.add2_mode == 3) {             // rare (1x1 stride-2 shortcut gradient): its address again, then the load
                                int64_t a2;
                                { RN_EPI_ADDR(GENERAL) a2 = a2off; (void)aoff; (void)off; }
                                if (a2 >= 0) { const float4 q = *reinterpret_cast<const float4 *>(add2 + a2); ad[0] += q.x; ad[1] += q.y; ad[2] += q.z; ad[3] += q.w; }
                            }
                            RN_EPI_FINISH()
                        }
                    }
                }
            }
        } else {
            for (int r = tid / CPR; r < RP; r += RPP) {
                const int64_t m = (int64_t)m0 + pass * RP + r;
                if (m >= M || !col_ok) break;
                float4 t = *reinterpret_cast<const float4 *>(T + r * LDT + 4 * c4);
                if constexpr (HALF) { const float ru = row_unscale[pass * RP + r]; t.x *= ru; t.y *= ru; t.z *= ru; t.w *= ru; }
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
    if (partial == nullptr && !rn_span) rn_amax_note(d.y_amax, m0 / HoWo, rn_am);
}




static inline int check_desc(const rn_conv_desc *d) {
    if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return RN_EINVAL;
    if (d->Cin < 4 || (d->Cin & 3)) return RN_EINVAL;                     // 16-byte chunks must not straddle taps
    if ((int64_t)d->Hi * d->Wi * d->Cin > 0x7fffffffLL) return RN_EINVAL; // in-image offsets are 32-bit
    {   // buffer addressing: the images one 256-row tile can touch, and the packed weights, within 2 GiB each
        const int64_t HoWo = (int64_t)d->Ho * d->Wo, span = 255 / HoWo + 2;
        if (d->x_batch_stride < 0 || (span - 1) * d->x_batch_stride * 4 + (int64_t)d->Hi * d->Wi * d->Cin * 4 > 0x7fffffffLL) return RN_EINVAL;
        const int64_t Kpad = ((int64_t)d->kh * d->kw * d->Cin + 31) / 32 * 32;
        if (d->Cout * Kpad * 4 > 0x7fffffffLL || (int64_t)d->N * HoWo > 0x7fffffffLL) return RN_EINVAL;
    }
    if (d->kh <= 0 || d->kw <= 0 || d->div_shift < 0 || d->div_shift > 2) return RN_EINVAL;
    if (d->add_mode < 0 || d->add_mode > 2 || d->act < 0 || d->act > 2) return RN_EINVAL;
    if (d->mask_mode < 0 || (d->mask_mode & ~(3 | RN_MASK_BITS)) || (d->mask_mode & 3) == 3 || d->mask_mode == RN_MASK_BITS) return RN_EINVAL;
    // sign bits (read: mask_mode | RN_MASK_BITS; written: sign_out) live at element offset >> 5: whole words per pixel and per image
    if (((d->mask_mode & RN_MASK_BITS) || d->sign_out != nullptr) && ((d->Cout & 31) || (d->y_batch_stride & 31))) return RN_EINVAL;
    if (d->os < 1 || d->oo_h < 0 || d->oo_w < 0 || (d->add2_mode != 0 && d->add2_mode != 3)) return RN_EINVAL;
    if ((d->Ho - 1) * d->os + d->oo_h >= d->Hy || (d->Wo - 1) * d->os + d->oo_w >= d->Wy) return RN_EINVAL;
    if (d->os != 1 && d->add_mode == 2) return RN_EINVAL;
    if (d->w_format < 0 || d->w_format > 3) return RN_EINVAL;
    if (d->w_format == 3 && (d->x_amax == nullptr || d->w_unscale == nullptr)) return RN_EINVAL;
    if (d->w_batch_stride < 0 || (d->w_batch_stride != 0 && ((int64_t)d->Ho * d->Wo) % 256 != 0)) return RN_EINVAL;
    return RN_OK;
}

