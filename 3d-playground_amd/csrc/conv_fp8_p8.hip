// fp8 (OCP e4m3fn) implicit-GEMM convolution, eight-wave 256 x 256 x 128 tile with a phased K loop (round 4) -- conv_bf16_p8.hip's
// structure for the inference form of BASELINE configs[4] (conv_fp8.hip: the fprop of every convolution of the detector behind the fp32
// stem, D/model.py:59-205, D/utils.py:12-80): the stride-1 same-size layers with Cin % 128 == 0.  Same rn_conv_desc, same e4m3 weights
// and per-channel scales, same epilogue arithmetic as conv_fp8.hip; its launchers choose between the two (fp8_p8_pick).
//
// What changes against the bf16 form: a staged row is again 128 bytes, now 128 K values; the MFMA is
// v_mfma_scale_f32_16x16x128_f8f6f4 (block scales 2^0: the scaled form is the one at the fp8 rate), one per 16 x 16 block and K-tile,
// 32 cycles each -- a K-tile is four phases of 8 MFMAs, the same ~1000 matrix-core cycles per wave between barriers at twice the math per
// staged byte.  A lane's operand is 32 consecutive K values of row lane & 15 (K quarter lane >> 4): two ds_read_b128, chunks 2 q and
// 2 q + 1 of the row's eight.  The bf16 kernel's chunk permutation is 2-way conflicting under that read; the one used here -- chunk c of
// row r in slot c ^ s(r), s(r) = bit 2 of r | bit 1 of r << 2 -- was found by enumerating the linear permutations over the real service
// groups of ds_read_b128 (MI355X_MICROARCH.md, LDS) and is conflict-free for both reads.
// Epilogue: two exchange levels (v_permlane16_swap, then v_permlane32_swap) give a lane SIXTEEN consecutive channels of one pixel --
// 16-byte e4m3 stores and addend loads, as in conv_fp8.hip.
// Roofline: MFMA (~5 PFLOP/s dense fp8).
#include <stdlib.h>

#include "common.h"

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int Q8_BM = 256, Q8_BN = 256, Q8_BK = 128, Q8_ROWB = 128;
constexpr int Q8_OPB = 256 * Q8_ROWB;            // bytes of one operand tile: 32 KB
constexpr int Q8_BUFB = 2 * Q8_OPB;              // one K-tile: pixels' rows, then weights' rows
constexpr int Q8_LDS = 2 * Q8_BUFB;              // 128 KB (dynamic)
#define Q8_MAX 448.0f                            // largest finite e4m3fn
__device__ __forceinline__ int q8_swz(int r) { return ((r >> 2) & 1) | (((r >> 1) & 1) << 2); }
__device__ __forceinline__ float q8_clamp(float a) { return fminf(fmaxf(a, -Q8_MAX), Q8_MAX); }
__device__ __forceinline__ int q8_pack4(float a, float b, float c, float d) {      // conv_fp8.hip: f8_pack4
    const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(q8_clamp(a), q8_clamp(b), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(q8_clamp(c), q8_clamp(d), lo, true);
}

#ifndef Q8_ABL
#define Q8_ABL 0                                 // knock-outs for profiles/ (bits): 1 no stores, 2 no addend loads, 4 no staging after a tile's first two K-tiles, 8 no MFMAs
#endif
struct Q8Tap { int r, s, c; };                   // filter row, filter column, first channel of a K-tile (wave-uniform)
struct Q8Args { float add_scale, out_inv_scale; };   // conv_fp8.hip: Fp8Args

// ---- epilogue: u = acc * scale[c] + shift[c]; u += add * add_scale; [ReLU]; y = e4m3(u * out_inv_scale) (conv_fp8.hip's arithmetic,
// with out_inv_scale -- positive -- folded into the three factors: the ReLU commutes with it; results identical on exact operands).
// acc[rb][cb][e]: pixel row rb * 16 + lr, channel cb * 16 + 4 lg + e of the wave's 128 x 64 (lr = lane & 15, lg = lane >> 4): a lane holds
// four consecutive channels of each of the four 16-channel blocks.  The arithmetic runs IN that layout, two channels per instruction
// (v_pk_fma_f32, v_cvt_pk_f32_fp8, v_cvt_pk_fp8_f32; v_med3_f32 is the ReLU and the e4m3 clamp in one), and only the packed e4m3 dwords
// change lanes: the four lanes of a pixel transpose their 4 x 4 dwords (lane L ends with dwords j = 0..3 of block L = 16 consecutive
// channels, one 16-byte store) in four instructions -- v_permlane16_swap on (P0,P1), (P2,P3), then v_permlane32_swap on (Q0,Q2), (Q1,Q3).
// The transposition is its own inverse: the addend, loaded as the 16 bytes of block `lg`, goes through the same four instructions to
// arrive as the lane's four channels of every block.  (The first version exchanged fp32 values, 24 instructions per 16 channels, and
// did every operation per channel: ~9 VALU instructions per element, 5 us per tile -- more than a 1x1 layer's K loop.)
__device__ __forceinline__ void q8_transpose4(unsigned (&p)[4]) {
    const u32x2 a = __builtin_amdgcn_permlane16_swap(p[0], p[1], false, false), b = __builtin_amdgcn_permlane16_swap(p[2], p[3], false, false);
    const u32x2 c = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false), e = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    p[0] = c[0]; p[1] = e[0]; p[2] = c[1]; p[3] = e[1];
}
template <bool RELU, bool ADD>
__device__ __forceinline__ void q8_epilogue(const f32x4 (&acc)[8][4], const rn_conv_desc &d, unsigned char *__restrict__ y,
                                            const float *__restrict__ scale, const float *__restrict__ shift,
                                            const unsigned char *__restrict__ add, const Q8Args qa, const int mw, const int nw, const int M,
                                            const int lane) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int lr = lane & 15, lg = lane >> 4;
    const int col_raw = nw + lg * 16;                            // the block this lane stores (and loads the addend of)
    const bool col_ok = col_raw < d.Cout;                        // Cout % 16 == 0 (launcher): a block is inside or outside
    const int col = col_ok ? col_raw : 0;                        // loads stay inside the tensors; nothing is stored
    const float is = qa.out_inv_scale;
    const f32x2 as2 = {qa.add_scale * is, qa.add_scale * is};
    f32x2 sc[4][2], sh[4][2];                                    // the lane's four channels of each block, scaled by 1 / out scale
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int c0 = nw + cb * 16 + 4 * lg;
        const bool ok = c0 < d.Cout;
        const float4 s4 = (ok && scale != nullptr) ? *reinterpret_cast<const float4 *>(scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 h4 = (ok && shift != nullptr) ? *reinterpret_cast<const float4 *>(shift + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        sc[cb][0] = f32x2{s4.x * is, s4.y * is}; sc[cb][1] = f32x2{s4.z * is, s4.w * is};
        sh[cb][0] = f32x2{h4.x * is, h4.y * is}; sh[cb][1] = f32x2{h4.z * is, h4.w * is};
    }
    // all eight pixel blocks' addends are requested before the first is used (a residual 1x1 layer's tile is mostly this round trip)
    i32x4 aqs[8];
    if constexpr (ADD) {
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            const int m = mw + rb * 16 + lr;
            aqs[rb] = *reinterpret_cast<const i32x4 *>(add + (int64_t)(m < M ? m : M - 1) * d.Cout + col);
        }
    }
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        const int m = mw + rb * 16 + lr;
        const bool row_ok = m < M;
        const int64_t off = (int64_t)(row_ok ? m : M - 1) * d.Cout + col;
        unsigned aq[4] = {0u, 0u, 0u, 0u};
        if constexpr (ADD) {
            if (!(Q8_ABL & 2)) {
                const i32x4 t = aqs[rb];
                aq[0] = (unsigned)t[0]; aq[1] = (unsigned)t[1]; aq[2] = (unsigned)t[2]; aq[3] = (unsigned)t[3];
            }
            q8_transpose4(aq);                                   // -> the lane's four channels of block cb in aq[cb]
        }
        unsigned o[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x2 u0 = f32x2{acc[rb][cb][0], acc[rb][cb][1]} * sc[cb][0] + sh[cb][0];
            f32x2 u1 = f32x2{acc[rb][cb][2], acc[rb][cb][3]} * sc[cb][1] + sh[cb][1];
            if constexpr (ADD) {
                u0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)aq[cb], false) * as2 + u0;
                u1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)aq[cb], true) * as2 + u1;
            }
            const float lo = RELU ? 0.f : -Q8_MAX;
            const int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(u0.x, lo, Q8_MAX), __builtin_amdgcn_fmed3f(u0.y, lo, Q8_MAX), 0, false);
            o[cb] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(u1.x, lo, Q8_MAX), __builtin_amdgcn_fmed3f(u1.y, lo, Q8_MAX), w0, true);
        }
        q8_transpose4(o);                                        // -> dwords 0..3 of block lg
        const i32x4 ov = {(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
        if ((Q8_ABL & 1) ? (ov[0] == 0x12345678 && row_ok && col_ok) : (row_ok && col_ok)) *reinterpret_cast<i32x4 *>(y + off) = ov;
    }
}

// HALF: Cin is a multiple of 64 only -- the two 64-channel halves of a 128-wide K-tile may belong to different filter taps (or the second
// lie past the end of the reduction: zero-filled on both sides).  A lane always stages the same half (its logical chunk >> 2), so the
// tap is a per-lane choice between two wave-uniform shifts.
// PERSIST: the workgroup computes tiles tile, tile + tile_step, ... < tile_end; the first two K-tiles of the NEXT tile are staged before
// the epilogue of the current one (after the last barrier of a K loop no wave reads LDS any more), so a tile's load latency and its
// epilogue overlap with its neighbours' -- what the short reductions (1x1 layers: one to eight K-tiles) are made of.
// ROWS: the general stride-a geometry (a strided layer, or an output plane that is not the input plane): the pixel rows of a tile are no
// flat sequence of the input, so the lane keeps one byte offset per pixel row (four) instead of one plus a uniform 64-row advance; taps
// stay uniform shifts.
template <bool HALF, bool PERSIST, bool ROWS>
__device__ __forceinline__ void q8_tile(const rn_conv_desc &d, const unsigned char *__restrict__ x, const unsigned char *__restrict__ w,
                                        unsigned char *__restrict__ y, const float *__restrict__ scale, const float *__restrict__ shift,
                                        const unsigned char *__restrict__ add, const Q8Args qa, int tile, const int tile_step,
                                        const int tile_end, char *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave >> 2, wc = wave & 3;     // rows wr * 128, columns wc * 64 of the tile
    const int ntn = (d.Cout + Q8_BN - 1) / Q8_BN;
    const int HoWo = d.Ho * d.Wo;
    const int M = d.N * HoWo;                    // < 2^31 - 256 (launcher)
    const int Cin = d.Cin, K = d.kh * d.kw * Cin, nkt = (K + Q8_BK - 1) / Q8_BK;

    // Pixels' descriptor: the flat [M][Cin] tensor from `halo` rows in front of the tile (the farthest a tap reaches back).
    const int ab = d.b < 0 ? -d.b : d.b;
    const int halo = ((d.p < 0 ? -d.p : d.p) + (d.kh - 1) * ab) * d.Wi + (d.p_w < 0 ? -d.p_w : d.p_w) + (d.kw - 1) * ab;
    const v4i32 rs_b = make_rsrc(w, (unsigned)((int64_t)d.Cout * K));
    const unsigned lds0 = lds_addr(lds);

    // ---- staging (conv_bf16_p8.hip): instruction i of this wave fills rows 64 (i & 3) + 8 wave + (lane >> 3) of the pixels (i < 4) or
    // the weights; the swizzle of the row (its bits 1, 2) does not depend on i
    const int row0 = 8 * wave + (lane >> 3);
    const int chunk = (lane & 7) ^ q8_swz(row0);
    const bool hi_lane = (chunk & 4) != 0;       // HALF: this lane stages the K-tile's second 64 channels
    // per-tile staging state (set_tile): the tile's origin, the pixels' descriptor, the lane's offsets and validity bits
    int m0 = 0, n0 = 0;
    v4i32 rs_a = rs_b;
    unsigned voff_a = 0, voff_b = 0;
    unsigned vrow[4] = {0u, 0u, 0u, 0u};         // ROWS: the byte offset of the lane's pixel row j (without tap)
    unsigned pk = 0;                             // bits 8 j + r: filter row r of pixel row j reads inside the image; 8 j + 4 + s: column s
    auto set_tile = [&](const int tl) {
        m0 = (tl / ntn) * Q8_BM;
        n0 = (tl % ntn) * Q8_BN;
        const int n_first = ROWS ? m0 / HoWo : 0;              // ROWS: the descriptor starts at the tile's first image
        if constexpr (ROWS) {
            const int64_t a_bytes = (int64_t)(d.N - n_first) * d.x_batch_stride;
            rs_a = make_rsrc(x + (int64_t)n_first * d.x_batch_stride, (unsigned)(a_bytes > 0x7FFFFFFF ? 0x7FFFFFFF : a_bytes));
        } else {
            const int base_row = m0 > halo ? m0 - halo : 0;
            const int64_t a_bytes = ((int64_t)M - base_row) * Cin;
            rs_a = make_rsrc(x + (int64_t)base_row * Cin, (unsigned)(a_bytes > 0x7FFFFFFF ? 0x7FFFFFFF : a_bytes));
            voff_a = (unsigned)((m0 - base_row + row0) * Cin + (HALF ? chunk & 3 : chunk) * 16);
        }
        voff_b = (unsigned)((n0 + row0) * K + chunk * 16);       // rows past Cout: past the descriptor's range
        pk = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + row0 + 64 * j;
            if (m < M) {
                const unsigned img = (unsigned)m / (unsigned)HoWo, rem = (unsigned)m - img * (unsigned)HoWo;
                const int oh = (int)(rem / (unsigned)d.Wo), ow = (int)(rem - (unsigned)oh * (unsigned)d.Wo);
                if constexpr (ROWS)
                    vrow[j] = (unsigned)(((int)img - n_first) * (int)d.x_batch_stride + (oh * d.a * d.Wi + ow * d.a) * Cin + (HALF ? chunk & 3 : chunk) * 16);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < d.kh && (unsigned)(oh * d.a + d.p + t * d.b) < (unsigned)d.Hi) pk |= 1u << (8 * j + t);
                    if (t < d.kw && (unsigned)(ow * d.a + d.p_w + t * d.b) < (unsigned)d.Wi) pk |= 1u << (8 * j + 4 + t);
                }
            }
        }
    };
    auto uni = [](const v4i32 r) {
        v4i32 o;
        o.x = __builtin_amdgcn_readfirstlane(r.x); o.y = __builtin_amdgcn_readfirstlane(r.y);
        o.z = __builtin_amdgcn_readfirstlane(r.z); o.w = __builtin_amdgcn_readfirstlane(r.w);
        return o;
    };
    // instruction i of this wave for K-tile kt (at tap tp).  live = false (past the last K-tile): every lane sends the out-of-range
    // offset -- zeros into a buffer nobody reads -- instead of a branch around the instruction: with branches in the K loop the compiler
    // sank the MFMAs of three phases below them (every operand live at once, nine accumulators spilled)
    auto next_half = [&](Q8Tap &tp) {                            // selects, not branches (see dma)
        const int c2 = tp.c + 64;
        const bool wc_ = c2 == Cin;
        const int s2 = tp.s + (wc_ ? 1 : 0);
        const bool ws_ = s2 == d.kw;
        tp.c = wc_ ? 0 : c2;
        tp.s = ws_ ? 0 : s2;
        tp.r += ws_ ? 1 : 0;
    };
    auto dma = [&](const int i, const Q8Tap &tp, const Q8Tap &th, const int kt, const int buf, const bool live) {   // th: the tap of the second half
        const unsigned dst = lds0 + (unsigned)(buf * Q8_BUFB + (wave_u + 8 * i) * 1024);
        const bool hi_live = live && kt * Q8_BK + 64 < K;        // HALF: the second half of the last K-tile may lie past K
        if (i < 4) {
            const unsigned va = ROWS ? vrow[i] : voff_a;
            const int sh = __builtin_amdgcn_readfirstlane(((d.p + tp.r * d.b) * d.Wi + d.p_w + tp.s * d.b + (ROWS ? 0 : 64 * i)) * Cin + tp.c);
            const unsigned ok = (pk >> (8 * i + (tp.r & 3))) & (pk >> (8 * i + 4 + (tp.s & 3))) & (live ? 1u : 0u);
            if constexpr (HALF) {
                const int sh_hi = __builtin_amdgcn_readfirstlane(((d.p + th.r * d.b) * d.Wi + d.p_w + th.s * d.b + (ROWS ? 0 : 64 * i)) * Cin + th.c);
                const unsigned ok_hi = (pk >> (8 * i + (th.r & 3))) & (pk >> (8 * i + 4 + (th.s & 3))) & (hi_live ? 1u : 0u);
                const unsigned okl = hi_lane ? ok_hi : ok;
                dma16(uni(rs_a), dst, okl ? va + (unsigned)(hi_lane ? sh_hi : sh) : 0x80000000u, 0u);
            } else {
                dma16(uni(rs_a), dst, ok ? va + (unsigned)sh : 0x80000000u, 0u);
            }
        } else {
            const bool okb = HALF ? (hi_lane ? hi_live : live) : live;
            dma16(uni(rs_b), dst, okb ? voff_b : 0x80000000u, (unsigned)__builtin_amdgcn_readfirstlane(live ? kt * Q8_BK + (i - 4) * 64 * K : 0));
        }
    };
    auto next_tap = [&](Q8Tap &tp) {                            // a whole K-tile further
        next_half(tp);
        next_half(tp);
    };

    // ---- fragments: block (16 rows) of this wave's 128 pixel rows / 64 weight rows; the lane's 32 K values = chunks 2 lg, 2 lg + 1
    const int lr = lane & 15, lg = lane >> 4;
    const int ra = wr * 128 + lr, rbb = wc * 64 + lr;
    int a_ad[2], b_ad[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        a_ad[e] = ra * Q8_ROWB + 16 * ((2 * lg + e) ^ q8_swz(ra));
        b_ad[e] = Q8_OPB + rbb * Q8_ROWB + 16 * ((2 * lg + e) ^ q8_swz(rbb));
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Operand registers: ALL weight fragments of the K-tile (B0 / B1: 32 rows each) and the pixel fragments in four groups of 32 rows
    // through two sets -- 64 registers.  (The bf16 kernel's scheme, two 64-row pixel sets and two weight sets = 96, does not fit here:
    // the 8-register operands of this MFMA fragment the file, and nine accumulators spilled.)
    i32x8 A[2][2], B0[2], B1[2];
    auto rd = [&](const char *S, const int (&ad)[2], int blk) {
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(S + ad[0] + blk * 2048), hi = *reinterpret_cast<const i32x4 *>(S + ad[1] + blk * 2048);
        return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto read_a = [&](i32x8 (&f)[2], const char *S, int g) {     // pixel rows 32 g .. 32 g + 31 of the wave's 128
        f[0] = rd(S, a_ad, 2 * g);
        f[1] = rd(S, a_ad, 2 * g + 1);
    };
    auto read_b = [&](i32x8 (&f)[2], const char *S, int q) {     // weight rows 32 q .. 32 q + 31 of the wave's 64
        f[0] = rd(S, b_ad, 2 * q);
        f[1] = rd(S, b_ad, 2 * q + 1);
    };
    const int one = 0x7F7F7F7F;                                  // E8M0 block scales: 2^0 in every byte
    // weight fragment first: a lane then holds 4 consecutive channels of one pixel
    auto mma = [&](const i32x8 (&fa)[2], const i32x8 (&fb)[2], int g, int q) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[2 * g + i][2 * q + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[2 * g + i][2 * q + j], 0, 0, 0, one, 0, one);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);                       // phases stay in this order (the reads of all of them in front of the
    };                                                           // MFMAs of all of them is what the compiler prefers)

    // ---- a tile's prologue: K-tile 0 -> buffer 0, K-tile 1 -> buffer 1 (all 8 instructions each)
    Q8Tap t2 = {0, 0, 0}, t2h = {0, 0, 0};
    auto stage_first_two = [&]() {
        t2 = Q8Tap{0, 0, 0};
        t2h = t2;
        next_half(t2h);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(i, t2, t2h, 0, 0, true);
        next_tap(t2);
        next_tap(t2h);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(i, t2, t2h, 1, 1, nkt > 1);
        next_tap(t2);
        next_tap(t2h);                                           // now the tap of K-tile t + 2 (in iteration t)
    };
    set_tile(tile);
    stage_first_two();
  for (;;) {
    // wait for K-tile 0 (a persistent workgroup: for everything, the last tile's stores included -- its loads were issued an epilogue ago)
    if (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    read_a(A[0], lds, 0);
    read_b(B0, lds, 0);

    // ---- main loop: K-tile t in buffer t & 1, eight phases of 4 MFMAs = (pixel group g, weight half): (0,0) (0,1) (1,0) (1,1) .. (3,0)
    // | barrier | (3,1).  A phase's operands are read one or two phases ahead; the last read of the buffer (group 3) is consumed by phase
    // (3,0), so at the barrier every read of it has returned and K-tile t + 1 has landed; phase (3,1) then reads the next K-tile's first
    // operands and restages the buffer for K-tile t + 2 between its MFMAs.
    for (int t = 0; t < nkt; ++t) {
        const char *S = lds + (t & 1) * Q8_BUFB;
        const char *Sn = lds + ((t + 1) & 1) * Q8_BUFB;
        const bool stage = t + 2 < nkt;
        read_b(B1, S, 1);
        mma(A[0], B0, 0, 0);
        read_a(A[1], S, 1);
        mma(A[0], B1, 0, 1);
        mma(A[1], B0, 1, 0);
        read_a(A[0], S, 2);
        mma(A[1], B1, 1, 1);
        mma(A[0], B0, 2, 0);
        read_a(A[1], S, 3);
        mma(A[0], B1, 2, 1);
        mma(A[1], B0, 3, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        read_b(B0, Sn, 0);                                       // after the last K-tile: stale bytes nobody uses (a conditional read costs
        read_a(A[0], Sn, 0);                                     // copies of the 8-register operands at the join)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int i = g >> 1, j = g & 1;
            acc[6 + i][2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(B1[j], A[1][i], acc[6 + i][2 + j], 0, 0, 0, one, 0, one);
            dma(2 * g, t2, t2h, t + 2, t & 1, stage);
            dma(2 * g + 1, t2, t2h, t + 2, t & 1, stage);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        next_tap(t2);
        if constexpr (HALF) next_tap(t2h);
    }

    // ---- the next tile's first two K-tiles go out before this tile's epilogue
    const int em0 = m0, en0 = n0;
    const int next = tile + tile_step;
    const bool has_next = PERSIST && next < tile_end;
    if (has_next) {
        set_tile(next);
        stage_first_two();
    }
    const bool relu = d.act == 1, has_add = d.add_mode == 1;
#define Q8_EPI(RELU, ADD) q8_epilogue<RELU, ADD>(acc, d, y, scale, shift, add, qa, em0 + wr * 128, en0 + wc * 64, M, lane)
    if (has_add) { if (relu) Q8_EPI(true, true); else Q8_EPI(false, true); }
    else { if (relu) Q8_EPI(true, false); else Q8_EPI(false, false); }
#undef Q8_EPI
    if (!has_next) break;
    tile = next;
  }
}

template <bool HALF, bool ROWS>
__global__ __launch_bounds__(512, 2) void conv_igemm_fp8_p8_kernel(const rn_conv_desc d, const unsigned char *__restrict__ x,
                                                                  const unsigned char *__restrict__ w, unsigned char *__restrict__ y,
                                                                  const float *__restrict__ scale, const float *__restrict__ shift,
                                                                  const unsigned char *__restrict__ add, const Q8Args qa) {
    extern __shared__ __attribute__((aligned(16))) char q8_lds[];
    q8_tile<HALF, false, ROWS>(d, x, w, y, scale, shift, add, qa, xcd_remap(blockIdx.x, gridDim.x), 0, 0, q8_lds);
}

// Persistent form: gridDim.x (a multiple of 8) workgroups share ntiles tiles.  The workgroups of an XCD (blockIdx & 7) own one contiguous
// range of the tiles (xcd_remap's partition) and walk it together: in pass i workgroup (x, slot) takes tile lo_x + i * gridDim.x / 8 + slot.
template <bool HALF, bool ROWS>
__global__ __launch_bounds__(512, 2) void conv_igemm_fp8_p8_persist_kernel(const rn_conv_desc d, const unsigned char *__restrict__ x,
                                                                          const unsigned char *__restrict__ w, unsigned char *__restrict__ y,
                                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                                          const unsigned char *__restrict__ add, const Q8Args qa, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char q8_lds[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, hi = lo + q + (xcd < r ? 1 : 0);
    if (lo + slot >= hi) return;
    q8_tile<HALF, true, ROWS>(d, x, w, y, scale, shift, add, qa, lo + slot, per, hi, q8_lds);
}

// Grouped launch (rn_conv_igemm_fp8_grouped): the pyramid levels of a head layer as ONE grid; a workgroup finds its problem by tile id.
template <bool HALF>
__global__ __launch_bounds__(512, 2) void conv_igemm_fp8_p8_grouped_kernel(const rn_conv_group g, const unsigned char *__restrict__ w,
                                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                                          const Q8Args qa) {
    extern __shared__ __attribute__((aligned(16))) char q8_lds[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int i = 0; i < RN_MAX_GROUP - 1; ++i) p += (i + 1 < g.n && tile >= g.tile_end[i]) ? 1 : 0;
    p = __builtin_amdgcn_readfirstlane(p);
    const int first = p > 0 ? g.tile_end[p - 1] : 0;
    q8_tile<HALF, false, false>(g.d[p], reinterpret_cast<const unsigned char *>(g.x[p]), w, reinterpret_cast<unsigned char *>(g.y[p]), scale, shift,
                         reinterpret_cast<const unsigned char *>(g.add[p]), qa, tile - first, 0, 0, q8_lds);
}

// ---------------------------------------------------------------------------------------------- host side (used by conv_fp8.hip)
// What the kernel can compute: a stride-1 convolution whose output plane is the input plane, batch-dense NHWC operands, Cin a multiple of
// 64 (half a K-tile), at most 4 x 4 taps, a dense e4m3 result, no sigmoid, no upsampled addend.
static inline bool q8_same_size(const rn_conv_desc *d) { return d->a == 1 && d->Hi == d->Ho && d->Wi == d->Wo; }
bool rn_fp8_p8_group_ok(const rn_conv_desc *d) { return q8_same_size(d); }
bool rn_fp8_p8_legal(const rn_conv_desc *d, int y_is_f32) {
    if (y_is_f32 || d->a < 1 || d->a > 2 || d->div_shift != 0 || d->act == 2) return false;
    if (d->Cin < 64 || (d->Cin & 63) || (d->Cout & 15) || d->kh > 4 || d->kw > 4) return false;
    const int64_t plane = (int64_t)d->Hi * d->Wi, oplane = (int64_t)d->Ho * d->Wo;
    if (d->x_batch_stride != plane * d->Cin || d->y_batch_stride != oplane * d->Cout) return false;
    if (!q8_same_size(d) && ((int64_t)d->N * plane * d->Cin > 0x7fffffffLL || (d->b < 0))) return false;   // general geometry: offsets from the tile's first image
    if (d->os != 1 || d->oo_h != 0 || d->oo_w != 0 || d->Hy != d->Ho || d->Wy != d->Wo || d->add_mode == 2) return false;
    if (d->add_mode == 1 && d->add_batch_stride != d->y_batch_stride) return false;
    const int64_t K = (int64_t)d->kh * d->kw * d->Cin, M = (int64_t)d->N * oplane;
    const int64_t ab = d->b < 0 ? -d->b : d->b;
    const int64_t halo = (llabs((long long)d->p) + (d->kh - 1) * ab) * d->Wi + llabs((long long)d->p_w) + (d->kw - 1) * ab;
    if (M + 256 > 0x7fffffffLL || (256 + 2 * halo + 64) * d->Cin > 0x7fffffffLL || ((int64_t)d->Cout + 256) * K > 0x7fffffffLL) return false;
    return true;
}
int rn_fp8_p8_launch(const rn_conv_desc *d, const void *x, const void *w, void *y, const float *scale, const float *shift, const void *add,
                     float add_scale, float out_inv_scale, hipStream_t stream) {
    static const hipError_t attr = [] {
        const void *ks[] = {(const void *)conv_igemm_fp8_p8_kernel<false, false>, (const void *)conv_igemm_fp8_p8_kernel<true, false>,
                            (const void *)conv_igemm_fp8_p8_kernel<false, true>, (const void *)conv_igemm_fp8_p8_kernel<true, true>,
                            (const void *)conv_igemm_fp8_p8_persist_kernel<false, false>, (const void *)conv_igemm_fp8_p8_persist_kernel<true, false>,
                            (const void *)conv_igemm_fp8_p8_persist_kernel<false, true>, (const void *)conv_igemm_fp8_p8_persist_kernel<true, true>};
        for (const void *k : ks) {
            const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, Q8_LDS);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }();
    if (attr != hipSuccess) return (int)attr;
    // persistent form: one workgroup per CU (a multiple of 8) once every workgroup has at least two tiles; RN_P8_PERSIST=0: never (A/B)
    static const int n_wg = [] {
        const char *e = getenv("RN_P8_PERSIST");
        if (e && atoi(e) == 0) return 0;
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
        return v / 8 * 8;
    }();
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    const int64_t tiles = ((M + 255) / 256) * ((d->Cout + 255) / 256);
    if (tiles > 0x7fffffff) return RN_EINVAL;
    Q8Args qa;
    qa.add_scale = add_scale;
    qa.out_inv_scale = out_inv_scale;
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(x), *wb = reinterpret_cast<const unsigned char *>(w);
    const unsigned char *ab = reinterpret_cast<const unsigned char *>(add);
    unsigned char *yb = reinterpret_cast<unsigned char *>(y);
    const bool half = (d->Cin & 127) != 0, rows = !q8_same_size(d);
    const dim3 blk(512);
#define Q8_GO(K, H, R, G, ...) hipLaunchKernelGGL((K<H, R>), dim3((unsigned)(G)), blk, Q8_LDS, stream, *d, xb, wb, yb, scale, shift, ab, qa, ##__VA_ARGS__)
    if (n_wg > 0 && tiles >= 2 * (int64_t)n_wg) {
        if (half) { if (rows) Q8_GO(conv_igemm_fp8_p8_persist_kernel, true, true, n_wg, (int)tiles); else Q8_GO(conv_igemm_fp8_p8_persist_kernel, true, false, n_wg, (int)tiles); }
        else { if (rows) Q8_GO(conv_igemm_fp8_p8_persist_kernel, false, true, n_wg, (int)tiles); else Q8_GO(conv_igemm_fp8_p8_persist_kernel, false, false, n_wg, (int)tiles); }
    } else {
        if (half) { if (rows) Q8_GO(conv_igemm_fp8_p8_kernel, true, true, tiles); else Q8_GO(conv_igemm_fp8_p8_kernel, true, false, tiles); }
        else { if (rows) Q8_GO(conv_igemm_fp8_p8_kernel, false, true, tiles); else Q8_GO(conv_igemm_fp8_p8_kernel, false, false, tiles); }
    }
#undef Q8_GO
    RN_LAUNCH_CHECK();
    return RN_OK;
}
int rn_fp8_p8_launch_grouped(const rn_conv_group *g, int tiles, const void *w, const float *scale, const float *shift, float add_scale,
                             float out_inv_scale, hipStream_t stream) {
    for (int i = 0; i < g->n; ++i)
        if (!q8_same_size(&g->d[i])) return RN_EINVAL;          // the grouped form has the same-size instances only (rn_fp8_p8_group_ok)
    static const hipError_t attr = [] {
        const hipError_t e = hipFuncSetAttribute((const void *)conv_igemm_fp8_p8_grouped_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, Q8_LDS);
        return e != hipSuccess ? e : hipFuncSetAttribute((const void *)conv_igemm_fp8_p8_grouped_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q8_LDS);
    }();
    if (attr != hipSuccess) return (int)attr;
    Q8Args qa;
    qa.add_scale = add_scale;
    qa.out_inv_scale = out_inv_scale;
    if (g->d[0].Cin & 127)
        hipLaunchKernelGGL(conv_igemm_fp8_p8_grouped_kernel<true>, dim3((unsigned)tiles), dim3(512), Q8_LDS, stream, *g,
                           reinterpret_cast<const unsigned char *>(w), scale, shift, qa);
    else
        hipLaunchKernelGGL(conv_igemm_fp8_p8_grouped_kernel<false>, dim3((unsigned)tiles), dim3(512), Q8_LDS, stream, *g,
                           reinterpret_cast<const unsigned char *>(w), scale, shift, qa);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
