// Split-operand implicit GEMM on v_mfma_f32_16x16x32_bf16: 128 x 128 output tile, four waves, each wave 32 tile rows x all 128
// columns (2 x 8 accumulators of 16 x 16), K-step 32, three workgroups per CU.  The activation operand never touches the LDS.
//
// Why this shape of instruction: the split kernels are not short of issue slots, the CHIP IS HOLDING ITS CLOCK DOWN under them.
// Stamped (tools/stamp_split.py, profiles/r03_split_clock.txt): inside the K loop of the 128 x 128 kernel (conv_igemm_tile.h,
// SPLIT 3, v_mfma_f32_32x32x16_bf16) the matrix pipe is ~77 % busy at a shader clock of 1.65-1.72 GHz, and the same kernel with
// one workgroup per CU runs at 2.28 GHz -- which is why knocking any part out of the K-step, or adding workgroups, returned so
// little (profiles/r03_split_knockout.txt).  What raises the clock is less energy per product (cdna_hip_programming.md 5.4 rule
// 28; MI355X_MICROARCH.md, DVFS give-back), and the 16x16x32 shape is such a lever: same cycles per FLOP, a higher clock --
// swapping only the instruction inside the 128 x 128 kernel (wrong results, timing only) moved the loop's clock 1717 -> 1877 MHz
// and the launch 1.54 -> 1.43 ms.
//
// Why this data path: a K = 32 instruction needs K-steps of 32, which double the LDS planes of the SPLIT 3 form.  Two forms that
// paid that price were built and dropped (profiles/r03_split_clock.txt, 4): 128 x 256 tiles with eight waves and 144 KB (one
// workgroup per CU: +3 % on K = 2304, -30 % on the Winograd GEMMs, nothing covers a tile's prologue and epilogue) and 128 x 128
// with single-buffered A planes and operands held in registers across a mid-step barrier (72 KB, two per CU, 202 registers: the
// loop's clock rose to 2.1 GHz but the MFMAs were 29 % of a wave's cycles).  This one keeps 48 KB and three per CU:
//   * the A operand of a 16x16x32 MFMA is, per lane, 8 consecutive channels of one pixel = 32 contiguous bytes of the NHWC tensor,
//     and with a wave owning its 32 rows alone no other wave needs them: each lane loads its own 2 x 8 values per K-step (two
//     buffer_load_dwordx4 per 16-row block, one step ahead, the range check as zero-fill for taps outside the image), splits them
//     in registers (split8) and feeds the MFMAs directly -- no A planes, no plane stores, no second barrier;
//   * only the weights are staged: pre-split (rn_split_weights: records of 96 bytes = h, m, l of 16 values), direct-to-LDS into two
//     buffers of 3 planes x 128 rows x 64 bytes, one barrier per 32 values of k;
//   * LDS planes are [rows][32 k] bf16 = 64-byte rows of four 16-byte chunks, chunk c of row r stored at slot c ^ 3 * ((r >> 3) & 1).
//     ds_read_b128 serves a wave in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
//     (MI355X_MICROARCH.md, LDS): a group holds every row r = lane & 15 once, rows 0-3 and 12-15 with one chunk and rows 4-11 with
//     the next, and with this permutation its 16 lanes cover the 16 slots of the 256-byte bank row once.  (The first layout,
//     c ^ ((r >> 2) & 3), is conflict-free for 8 CONSECUTIVE lanes, which is not how the hardware groups them: SQ_LDS_BANK_CONFLICT
//     was half of the LDS cycles.)
//   * the epilogue is wave-private: a wave transposes its accumulators through a 16-row strip of LDS of its own and stores whole
//     512-byte rows; no workgroup barrier after the K loop.
// 132 registers.  Conditions (the launcher's): pre-split weights, Cin a multiple of 32, div_shift 0, kh * kw <= 24, Cout a multiple
// of 4 and > 64 (a 128 x 64 instance for narrower layers measured neutral in round 3 and was removed in round 5), no input ReLU.
//
// TERMS = 2 (round 5, RN_FP32_SPLIT3; mfma_split.h, second half): the same kernel on v_mfma_f32_16x16x32_f16 with TWO fp16 terms per
// operand and THREE MFMAs per product.  The weight image has two planes (rn_split_weights_f16: 64-byte records, the row's own
// power-of-two scale), the activation values are multiplied by the tensor's power-of-two scale (from its amax word, one scalar load per
// workgroup) inside the split, and the epilogue multiplies every column by the two inverse scales (folded into the batch-norm scale it
// multiplies by anyway; one extra multiplication per element in the plain-GEMM form).  32 KB of weight planes instead of 48.
//
// STG (round 5, second half; TERMS 2, launches with K >= 256): the activation operand IS staged after all -- not for the LDS's sake but for the
// vector-memory pipe's, which pays per cache line a quarter-wave touches: the operand layout read straight from memory (16 consecutive lanes =
// 16 rows) runs at 9.2 TB/s chip-wide even out of the L2, row-coalesced loads at 31 (tools/probes/a_pattern_probe.hip,
// profiles/r05_mf16_bounds.txt).  A wave fetches its own 32 rows x 128 bytes of the K-step by four direct-to-LDS instructions of 8 rows each
// into a 4 KB strip nobody else touches (no extra barrier: its own s_waitcnt orders it) and reads them back as operands, conflict-free
// (conv_wgrad_geom.h: Mf16AGeom).  The rows' geometry -- the divisions -- is computed once per row into a table instead of once per reader.
// WV = 8 (same half of the round): the staged form with EIGHT waves on a 256 x 128 tile -- every wave still owns 32 rows x 128 columns, the
// weights' planes are shared by twice as many rows (0.75 x the bytes through the CU's memory pipe per product), the epilogue strips are 8 rows
// instead of 16 so that two workgroups (sixteen waves) fit a CU: 70 KB of LDS, 114-120 registers.  Every staged launch but the Winograd stage's.
#include <type_traits>

#include "conv_igemm_tile.h"
#include "conv_wgrad_geom.h"

typedef float f32x4a __attribute__((ext_vector_type(4)));

#ifndef RN_MF16_STG
#define RN_MF16_STG 1            // the fp16 form stages its activations through the LDS (row-coalesced loads); 0: straight into registers (A/B)
#endif
#ifndef RN_MF16_PF
#define RN_MF16_PF 1             // register sets of activation values in flight (A/B; RN_EXPERIMENT builds)
#endif
#ifndef RN_MF16_KO
#define RN_MF16_KO 0             // knock-outs (RN_EXPERIMENT builds; timing only, wrong results): 1 no split / MFMAs, 2 no activation loads,
#endif                           // 4 no result stores, 8 no weight DMA

// acc += a * b for one 16 x 16 tile and 32 values of k: the six products, smallest first (mfma_split.h: RN_SPLIT_MFMA)
#define RN_SPLIT_MFMA16(ACC, A, B)                                                        \
    do {                                                                                  \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A).l, (B).h, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A).h, (B).l, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A).m, (B).m, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A).m, (B).h, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A).h, (B).m, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((A).h, (B).h, ACC, 0, 0, 0);        \
    } while (0)

template <bool GENERAL, bool RAW, int TERMS, bool STG, int WV = 4>
__device__ __forceinline__ void conv_mf16_tile(const rn_conv_desc &d, const float *__restrict__ x, const float *__restrict__ w,
                                               float *__restrict__ y, const float *__restrict__ scale,
                                               const float *__restrict__ shift, const float *__restrict__ add,
                                               const float *__restrict__ mask, const float *__restrict__ add2, const int tile) {
    constexpr int BK = 32, BM = 32 * WV, BN = 128, NSN = BN / 16;    // WV waves of 32 rows x 128 columns each (4: 128 x 128; 8: 256 x 128)
    constexpr int SR = WV == 8 ? 8 : 16;                   // rows of a wave's epilogue strip (eight waves: half a 16-row block at a time)
    constexpr bool HALF = TERMS == 2;                      // two fp16 terms / three MFMAs instead of three bf16 terms / six
    constexpr int REC = 32 * TERMS, EB = 2 * TERMS;        // bytes of a pre-split record (16 values) / per weight element
    constexpr int BPL = BN * 16;                           // floats' worth of one 16-bit plane: rows x 64 bytes
    constexpr int BSTEP = TERMS * BPL;                     // one buffer: B planes h, m, l (TERMS 3) or hi, lo (TERMS 2)
    constexpr int NBI = TERMS * BN / 16, IB = NBI / WV;    // direct-to-LDS instructions (16 rows x 64 bytes each) per step / per wave
    constexpr int LDT = BN + 4;
    constexpr int LDSF = 2 * BSTEP > WV * SR * LDT ? 2 * BSTEP : WV * SR * LDT;   // two staging buffers; the epilogue's strips, one per wave
    static_assert(NBI % WV == 0 && (WV == 4 || WV == 8) && (TERMS == 2 || TERMS == 3), "tile shape");
    __shared__ float lds[LDSF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;              // a lane's row / column within a 16 x 16 tile, its block of 8 k values
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ntn = (d.Cout + BN - 1) / BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int HoWo = d.Ho * d.Wo;
    const int64_t M = (int64_t)d.N * HoWo;
    const int K = d.kh * d.kw * d.Cin;
    const int Kpad = (K + 31) / 32 * 32;
    const int nks = Kpad / BK;

    // ---- buffer descriptors: activations from the first image the tile touches, the pre-split weights whole (+ per-image offset)
    const int n_first = (int)(m0 / HoWo);
    const int64_t x_floats = ((int64_t)d.N - 1 - n_first) * d.x_batch_stride + (int64_t)d.Hi * d.Wi * d.Cin;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(x + (int64_t)n_first * d.x_batch_stride), (short)0,
        (int)(unsigned)(x_floats * 4 > 0x7FFFFFFF ? 0x7FFFFFFF : x_floats * 4), 0x00020000);
    const v4i32 rs_b = make_rsrc(reinterpret_cast<const char *>(w) + (int64_t)n_first * d.w_batch_stride * EB,
                                 (unsigned)((int64_t)d.Cout * Kpad * EB));
    // TERMS 2: every activation row has the power-of-two scale of ITS image (rn_conv_desc.x_amax: per image, or per row for the Winograd
    // stage), taken from the amax words below; the inverses wait in a 128-float table for the epilogue, which multiplies row by row.
    __shared__ float row_unscale[HALF ? BM : 1];
    float a_scale[2] = {1.f, 1.f};

    // ---- this lane's two activation rows (tile rows 32 * wave + 16 * sm + lr): tap mask and base offset of its 8 k values
    unsigned a_mask[2];
    int a_base[2];
    int a_n[2] = {-1, -1}, a_rem[2] = {0, 0};              // TERMS 2: the rows' images (absolute) and pixels; -1: a row past M
    // STG: the wave's 32 rows x 32 values of a K-step are STAGED -- row-coalesced direct-to-LDS loads into a 4 KB strip of the wave's own
    // (conv_wgrad_geom.h: Mf16AGeom), read back in the operand layout.  A lane fills rows 8 j + (lane >> 3), j = 0 .. 3.
    __shared__ __attribute__((aligned(16))) char a_stage[STG ? WV * Mf16AGeom::WAVE_BYTES : 16];
    __shared__ int4 row_tab[STG ? BM : 1];                 // per tile row: (byte offset of its pixel at tap (0, 0), tap mask, image, pixel)
    unsigned s_mask[4] = {0u, 0u, 0u, 0u};
    int s_base[4] = {0, 0, 0, 0};
    auto row_geom = [&](int row, unsigned &mk, int &base, int &n_abs, int &rem_) {
        int a_h = -(1 << 28), a_w = 0, a_img = 0;
        n_abs = -1; rem_ = 0;
        if ((int64_t)m0 + row < M) {
            const unsigned rel = (unsigned)(m0 - n_first * HoWo + row);
            const unsigned n = rel / (unsigned)HoWo;
            const unsigned rem = rel - n * (unsigned)HoWo;
            const unsigned oh = rem / (unsigned)d.Wo, ow = rem - oh * (unsigned)d.Wo;
            a_img = (int)((int64_t)n * d.x_batch_stride * 4);
            a_h = (int)oh * d.a + d.p;
            a_w = (int)ow * d.a + d.p_w;
            n_abs = n_first + (int)n; rem_ = (int)rem;
        }
        mk = 0;
        for (int r = 0, t = 0; r < d.kh; ++r)
            for (int s_ = 0; s_ < d.kw; ++s_, ++t) {
                const int ih = a_h + r * d.b, iw = a_w + s_ * d.b;
                mk |= (unsigned)(((ih | iw) >= 0) & (ih < d.Hi) & (iw < d.Wi)) << t;
            }
        base = a_img + (a_h * d.Wi + a_w) * d.Cin * 4;
    };
    if constexpr (STG) {
        if (tid < BM) {                                    // one thread per tile row: the divisions once, not once per reader
            unsigned mk; int base, n_abs, rem_;
            row_geom(tid, mk, base, n_abs, rem_);
            row_tab[tid] = make_int4(base, (int)mk, n_abs, rem_);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int4 e = row_tab[32 * wave + Mf16AGeom::dma_row(lane, j)];
            s_base[j] = e.x + 16 * Mf16AGeom::dma_chunk(lane, j);
            s_mask[j] = (unsigned)e.y;
        }
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const int4 e = row_tab[32 * wave + 16 * sm + lr];
            a_n[sm] = e.z; a_rem[sm] = e.w;
            a_mask[sm] = 0u; a_base[sm] = 0;
        }
    } else {
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            row_geom(32 * wave + 16 * sm + lr, a_mask[sm], a_base[sm], a_n[sm], a_rem[sm]);
            a_base[sm] += 8 * lg * 4;
        }
    }
    int f_r = 0, f_s = 0, f_c = 0;                         // tap and channel offset of the next step to load
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    struct ARegs { f32x4v v[4]; };
    auto load_a = [&](ARegs &ar) {
#if RN_MF16_KO & 2
        if (f_c >= 0) { f_c += BK; return; }
#endif
        const int t = f_r * d.kw + f_s;
        const int delta = (f_r * d.Wi + f_s) * d.b * d.Cin * 4;
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const unsigned valid = 0u - ((a_mask[sm] >> (t & 31)) & 1u);
            const unsigned v = ((unsigned)(a_base[sm] + delta) & valid) | (0x80000000u & ~valid);
            ar.v[2 * sm] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)v, f_c * 4, 0));
            ar.v[2 * sm + 1] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)(v + 16u), f_c * 4, 0));
        }
        f_c += BK;
        const bool wrap = f_c >= d.Cin;
        f_c = wrap ? 0 : f_c;
        f_s += wrap ? 1 : 0;
        const bool wrap_s = f_s == d.kw;
        f_s = wrap_s ? 0 : f_s;
        f_r += wrap_s ? 1 : 0;
    };

    // STG: the K-step's 32 values of the wave's 32 rows -> its strip (four instructions of 8 rows x 128 bytes), and back as operands
    const v4i32 rs_a4 = make_rsrc(x + (int64_t)n_first * d.x_batch_stride, (unsigned)(x_floats * 4 > 0x7FFFFFFF ? 0x7FFFFFFF : x_floats * 4));
    const unsigned a_lds0 = lds_addr(a_stage) + (unsigned)__builtin_amdgcn_readfirstlane(wave) * Mf16AGeom::WAVE_BYTES;
    auto dma_a = [&]() {
#if RN_MF16_KO & 2
        if (f_c >= 0) { f_c += BK; return; }
#endif
        const int t = f_r * d.kw + f_s;
        const int delta = (f_r * d.Wi + f_s) * d.b * d.Cin * 4;
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(f_c * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned valid = 0u - ((s_mask[j] >> (t & 31)) & 1u);
            dma16(rs_a4, a_lds0 + 1024u * j, ((unsigned)(s_base[j] + delta) & valid) | (0x80000000u & ~valid), soff);
        }
        f_c += BK;
        const bool wrap = f_c >= d.Cin;
        f_c = wrap ? 0 : f_c;
        f_s += wrap ? 1 : 0;
        const bool wrap_s = f_s == d.kw;
        f_s = wrap_s ? 0 : f_s;
        f_r += wrap_s ? 1 : 0;
    };
    const char *a_rd = a_stage + wave * Mf16AGeom::WAVE_BYTES;
    const int a_ra[2][2] = {{Mf16AGeom::read_addr(lane, 0, 0), Mf16AGeom::read_addr(lane, 0, 1)},
                            {Mf16AGeom::read_addr(lane, 1, 0), Mf16AGeom::read_addr(lane, 1, 1)}};
    auto read_a = [&](ARegs &ar) {
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            ar.v[2 * sm] = *reinterpret_cast<const f32x4v *>(a_rd + a_ra[sm][0]);
            ar.v[2 * sm + 1] = *reinterpret_cast<const f32x4v *>(a_rd + a_ra[sm][1]);
        }
    };

    // ---- weight planes: instruction q of the workgroup fills 16 rows of one plane; lane -> row (lane >> 2), slot (lane & 3).
    // The slot holds chunk c = slot ^ 3 * ((row >> 3) & 1) = (lane & 3) ^ 3 * ((lane >> 5) & 1): k values 8c .. 8c+7 of the step, i.e. bytes
    // (c & 1) * 16 of the plane's half-record in 16-value record c >> 1.
    unsigned b_voff[IB];
#pragma unroll
    for (int j = 0; j < IB; ++j) {
        const int q = wave * IB + j, plane = q / (BN / 16), brow = (q % (BN / 16)) * 16 + (lane >> 2);
        const int c = Mf16Geom::dma_chunk(lane);
        const int n = n0 + brow;
        b_voff[j] = n < d.Cout ? (unsigned)(n * Kpad * EB + (c >> 1) * REC + plane * 32 + (c & 1) * 16) : 0x80000000u;
    }
    const unsigned lds0 = lds_addr(lds);
    auto dma_b = [&](int ks, int buf) {
#if RN_MF16_KO & 8
        if (ks > 0) return;
#endif
#pragma unroll
        for (int j = 0; j < IB; ++j)
            dma16(rs_b, lds0 + (unsigned)(buf * BSTEP * 4 + (wave_u * IB + j) * 1024), b_voff[j], (unsigned)(ks * 2 * REC));
    };

    f32x4a acc[2][NSN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NSN; ++j) acc[i][j] = f32x4a{0.f, 0.f, 0.f, 0.f};
    // plane 0 of this lane's operand chunk of weight rows 16 * t + lr (floats, within a buffer): 16 * row + 4 * (lg ^ 3 * ((row >> 3) & 1));
    // 16 * t does not change (row >> 3) & 1, so one address and an immediate per t
    const int fb0 = Mf16Geom::read_addr(lane, 0) / 4;

    // ---- K loop.  `cur` holds the A values of step ks (loaded during step ks - 1): split into the MFMA operands, then the same
    // registers receive step ks + 1 -- a whole MFMA phase to arrive.  Unconditional (one basic block): past the last step the loads
    // return zeros / stale weights nobody uses, and all have landed (vmcnt(0)) before the epilogue reuses the LDS.
    constexpr int PF = RN_MF16_PF;                         // K-steps of activation values in flight (register sets)
    static_assert(!STG || PF == 1, "staged activations: one step in flight");
    ARegs ar[PF] = {};
    if constexpr (STG) dma_a(); else load_a(ar[0]);
    dma_b(0, 0);
#pragma unroll
    for (int i = 1; i < PF; ++i) load_a(ar[i]);
    if constexpr (HALF) {
        // the rows' scales, while the first operands travel: per image from its exponent table (one 256-byte read per wave and image --
        // a wave's 32 rows lie in one image, rarely two), or per row from the plain words the Winograd input transform wrote
        int e[2];
        if (d.x_amax_row_stride == 0) {
            const unsigned char *tb = reinterpret_cast<const unsigned char *>(d.x_amax);
            e[0] = rn_amax_exp_lanes(a_n[0] < 0 ? nullptr : tb + (int64_t)a_n[0] * d.x_amax_img_stride * RN_AMAX_BYTES);
            e[1] = __ballot(a_n[1] != a_n[0]) == 0ull ? e[0]
                 : rn_amax_exp_lanes(a_n[1] < 0 ? nullptr : tb + (int64_t)a_n[1] * d.x_amax_img_stride * RN_AMAX_BYTES);
        } else {
#pragma unroll
            for (int sm = 0; sm < 2; ++sm)
                e[sm] = a_n[sm] < 0 ? 0 : (int)((reinterpret_cast<const unsigned *>(d.x_amax)[(int64_t)a_rem[sm] * d.x_amax_row_stride] >> 23) & 0xffu);
        }
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const int se = rn_f16_scale_exp_of(e[sm]);
            a_scale[sm] = rn_exp_to_float(se);
            if (lg == 0) row_unscale[32 * wave + 16 * sm + lr] = rn_exp_to_float(254 - se);      // (the wave's own rows: no other wave reads them)
        }
    }
    constexpr int CPR = BN / 4, RPI = 64 / CPR;              // 16-byte chunks per row, rows one wave instruction covers
    const int c4 = lane % CPR;
    const int col = n0 + 4 * c4;
    const bool col_ok = col < d.Cout;
    const bool vec = true;                                   // the launcher sends Cout % 4 == 0 only
    const int ncol = 4;
    rn_wait_but<4 * (PF - 1)>();
    __syncthreads();
    auto k_step = [&](int ks, int rb, ARegs &cur) {
        if constexpr (STG) read_a(cur);                     // step ks landed in the strip (the wait that ended the previous step)
        asm volatile("" : "+v"(cur.v[0]), "+v"(cur.v[1]), "+v"(cur.v[2]), "+v"(cur.v[3]));   // the compiler's wait for `cur` here, where it is free
        typename std::conditional<HALF, SplitH8, Split8>::type sa[2];
#if RN_MF16_KO & 1
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q & 1][q >> 1] += cur.v[q];
        dma_b(ks + 1, rb ^ 1);
        if constexpr (STG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dma_a(); } else load_a(cur);
        rn_wait_dma();
        __syncthreads();
        if (ks >= 0) return;
#endif
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const float av[8] = {cur.v[2 * sm][0], cur.v[2 * sm][1], cur.v[2 * sm][2], cur.v[2 * sm][3],
                                 cur.v[2 * sm + 1][0], cur.v[2 * sm + 1][1], cur.v[2 * sm + 1][2], cur.v[2 * sm + 1][3]};
            if constexpr (HALF) sa[sm] = split8h(av, a_scale[sm]);
            else sa[sm] = split8(av);
        }
        dma_b(ks + 1, rb ^ 1);
        if constexpr (STG) {
            // the strip is this wave's own: its reads above have returned (the split consumed them; the explicit wait is for the asm
            // below, which the compiler does not order against them), so step ks + 1 may land on top
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_a();
        } else load_a(cur);
        const float *S = lds + rb * BSTEP + fb0;
#pragma unroll
        for (int sn = 0; sn < NSN; ++sn) {
            if constexpr (HALF) {
                SplitH8 sb;
                sb.h = *reinterpret_cast<const f16x8 *>(S + 256 * sn);
                sb.l = *reinterpret_cast<const f16x8 *>(S + 256 * sn + BPL);
#pragma unroll
                for (int sm = 0; sm < 2; ++sm) RN_SPLITH_MFMA16(acc[sm][sn], sa[sm], sb);
            } else {
                Split8 sb;
                sb.h = *reinterpret_cast<const bf16x8 *>(S + 256 * sn);
                sb.m = *reinterpret_cast<const bf16x8 *>(S + 256 * sn + BPL);
                sb.l = *reinterpret_cast<const bf16x8 *>(S + 256 * sn + 2 * BPL);
#pragma unroll
                for (int sm = 0; sm < 2; ++sm) RN_SPLIT_MFMA16(acc[sm][sn], sa[sm], sb);
            }
        }
        RN_PIN();
        // B planes of step ks + 1 landed, A registers of step ks + 1 arrived: everything but the newest PF - 1 sets of four loads
        // (the memory pipe returns in order, and the four loads of step ks + PF were issued last)
        rn_wait_but<4 * (PF - 1)>();
        __syncthreads();
    };
    constexpr int UN = PF > 2 ? PF : 2;
    for (int ks = 0; ks < nks; ks += UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (u == 0 || ks + u < nks) k_step(ks + u, u & 1, ar[u % PF]);
    }

    // ---- epilogue, WAVE-PRIVATE: a wave owns tile rows 32 w .. 32 w + 31 and all 128 columns, so it transposes its accumulators
    // through a 16-row strip of LDS of its own (the staging buffers are free after the last barrier) and nobody waits for anybody:
    // no workgroup barrier after the K loop.  Accumulator element e of lane l is row 4 * (l >> 4) + e, column l & 15 of its
    // 16 x 16 tile; out of the strip a lane takes float4s: BN / 4 consecutive lanes one whole row segment, so out, add and mask all
    // move as 16-byte accesses (conv_igemm_tile.h: same arithmetic, same macros).
    float *T = lds + wave * (SR * LDT);
    // (requesting these factors BEFORE the K loop, to have their latency covered, costs 8 % of the family's time -- 29.8 -> 32.3 ms per
    // training step, tools/dbg/ab_lib.sh -- they stay here)
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!RAW && col_ok && scale != nullptr) sc[j] = scale[col + j];
        if (!RAW && col_ok && shift != nullptr) sh[j] = shift[col + j];
        // TERMS 2: the accumulators hold the product of the SCALED operands: the weight row's inverse scale (a power of two) rides in
        // the column's factor, the activation row's is applied row by row below
        if (HALF && col_ok) sc[j] *= d.w_unscale[(d.w_batch_stride != 0 ? (int64_t)n_first * d.Cout : 0) + col + j];
    }
    float rn_am = 0.f;                                       // largest |y| this lane stored (rn_conv_desc.y_amax)
    const int64_t m_last = (int64_t)m0 + BM - 1 < M ? (int64_t)m0 + BM - 1 : M - 1;
    const bool rn_span = (int)(m0 / HoWo) != (int)(m_last / HoWo);           // the tile's rows lie in more than one image (scalar)
#pragma unroll
    for (int smh = 0; smh < 2 * (16 / SR); ++smh) {
        const int sm = smh / (16 / SR), hp = smh % (16 / SR);   // the 16-row block and which SR rows of it go through the strip now
        if (SR == 16 || (lg >> 1) == hp) {
#pragma unroll
            for (int sn = 0; sn < NSN; ++sn)
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(4 * lg + e - hp * SR) * LDT + 16 * sn + lr] = acc[sm][sn][e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the strip is this wave's own: order within the wave is all it needs
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int NIT = SR / RPI, G = 4;                 // SR rows, RPI per instruction
        if (col_ok) {
#pragma unroll 1
            for (int g = 0; g < NIT; g += G) {
                int64_t off_[G];
                float4 mk_[G], ad_[G];
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int64_t mr = (int64_t)m0 + 32 * wave + 16 * sm + hp * SR + lane / CPR + RPI * (g + i);
                    const int64_t m = mr < M ? mr : M - 1;
                    RN_EPI_ADDR(GENERAL)
                    off_[i] = off;
                    mk_[i] = make_float4(1.f, 1.f, 1.f, 1.f);
                    ad_[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!RAW && d.mask_mode != 0) mk_[i] = rn_mask_load4(mask, off, (d.mask_mode & RN_MASK_BITS) != 0);
                    if (!RAW && d.add_mode != 0) ad_[i] = *reinterpret_cast<const float4 *>(add + aoff);
                    (void)a2off;
                }
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int r = lane / CPR + RPI * (g + i);
                    const int64_t m = (int64_t)m0 + 32 * wave + 16 * sm + hp * SR + r;
                    if (m < M) {
                        float4 t = *reinterpret_cast<const float4 *>(T + r * LDT + 4 * c4);
                        if constexpr (HALF) { const float ru = row_unscale[32 * wave + 16 * sm + hp * SR + r]; t.x *= ru; t.y *= ru; t.z *= ru; t.w *= ru; }
                        const int64_t off = off_[i];
                        float mk[4] = {mk_[i].x, mk_[i].y, mk_[i].z, mk_[i].w}, ad[4] = {ad_[i].x, ad_[i].y, ad_[i].z, ad_[i].w};
                        if (!RAW && d.add2_mode == 3) {
                            int64_t a2;
                            { RN_EPI_ADDR(GENERAL) a2 = a2off; (void)aoff; (void)off; }
                            if (a2 >= 0) { const float4 q = *reinterpret_cast<const float4 *>(add2 + a2); ad[0] += q.x; ad[1] += q.y; ad[2] += q.z; ad[3] += q.w; }
                        }
#if RN_MF16_KO & 4
                        if (t.x != 1.2345f) continue;
#endif
                        if (RAW) *reinterpret_cast<float4 *>(y + off) = HALF ? make_float4(t.x * sc[0], t.y * sc[1], t.z * sc[2], t.w * sc[3]) : t;
                        else { RN_EPI_FINISH() }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the strip's reads before the next half's writes
        __builtin_amdgcn_wave_barrier();
    }
    if (!RAW && !rn_span) rn_amax_note(d.y_amax, m0 / HoWo, rn_am);
}

#ifndef RN_MF16H_OCC
#define RN_MF16H_OCC 3           // workgroups per CU the fp16 form's registers are cut for (120 registers, 33 KB of LDS: 4 fit; A/B below)
#endif
template <bool GENERAL, bool RAW, int TERMS, bool STG, int WV = 4>
__global__ __launch_bounds__(64 * WV, WV == 8 ? 2 : (TERMS == 2 ? RN_MF16H_OCC : 3)) void conv_igemm_mf16_kernel(const rn_conv_desc d, const float *__restrict__ x,
                                                              const float *__restrict__ w, float *__restrict__ y,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              const float *__restrict__ add, const float *__restrict__ mask,
                                                              const float *__restrict__ add2) {
    conv_mf16_tile<GENERAL, RAW, TERMS, STG, WV>(d, x, w, y, scale, shift, add, mask, add2, xcd_remap(blockIdx.x, gridDim.x));
}

template <int TERMS>
__global__ __launch_bounds__(256, TERMS == 2 ? RN_MF16H_OCC : 3) void conv_igemm_mf16_grouped_kernel(const rn_conv_group g, const float *__restrict__ w,
                                                                      const float *__restrict__ scale,
                                                                      const float *__restrict__ shift) {
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int i = 0; i < RN_MAX_GROUP - 1; ++i) p += (i + 1 < g.n && tile >= g.tile_end[i]) ? 1 : 0;
    rn_conv_desc d = g.d[0];
    const float *x = g.x[0], *add = g.add[0], *mask = g.mask[0];
    float *y = g.y[0];
    int first = 0;
#pragma unroll
    for (int i = 1; i < RN_MAX_GROUP; ++i)
        if (p == i) { d = g.d[i]; x = g.x[i]; y = g.y[i]; add = g.add[i]; mask = g.mask[i]; first = g.tile_end[i - 1]; }
    conv_mf16_tile<true, false, TERMS, TERMS == 2 && RN_MF16_STG != 0>(d, x, w, y, scale, shift, add, mask, nullptr, tile - first);
}

// Which problems take this kernel: RN_OPT_MF16 = 0 turns it off (A/B); launches with fewer than RN_OPT_MF16_MIN tiles keep the
// 32x32x16 kernels.  (rn_get_option: the environment is read once, not per launch; the parity tests switch with rn_set_option.)
static int mf16_on() { return rn_get_option(RN_OPT_MF16); }
static int mf16_min_tiles() { return rn_get_option(RN_OPT_MF16_MIN); }
static bool mf16_geom_ok(const rn_conv_desc *d) {
    return (d->Cin % 32) == 0 && d->div_shift == 0 && d->kh * d->kw <= 24 && (d->Cout % 4) == 0 && d->Cout > 64 && !d->in_relu;
}
static bool mf16_ok(const rn_conv_desc *d) {
    if (!mf16_on() || !mf16_geom_ok(d)) return false;
    if (d->w_format == 3) return d->x_amax != nullptr && d->w_unscale != nullptr;
    return d->w_format == 1;
}
static int64_t mf16_tiles(const rn_conv_desc *d) {
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    return ((M + 127) / 128) * ((d->Cout + 127) / 128);
}

// include/retinanet_mi355x.h: does this problem run on an fp16-split kernel in RN_FP32_SPLIT3 mode?  Since the 128 x 128 / 256 x 64 tiles
// of conv_igemm_tile.h have the two-term form too (round 5, second half): every problem the split kernels take -- a reduction of at
// least rn_fp32_split_min_k() -- whatever its geometry.  (The split-K form, chosen by rn_conv_splitk_workspace_bytes, stays on the fp32 MFMA.)
extern "C" int rn_conv_igemm_wants_f16(const rn_conv_desc *d) {
    return rn_get_fp32_mfma() == RN_FP32_SPLIT3 && d->w_batch_stride >= 0;
}

// -> true if launched.  variant as rn_igemm_split_launch: 0 raw, 4 / 5 wide dense / general.
bool rn_igemm_mf16_launch(int variant, const rn_conv_desc *d, const float *x, const float *w, float *y, const float *scale,
                          const float *shift, const float *add, const float *mask, const float *add2, hipStream_t s, int *rc) {
    if (!mf16_ok(d) || (variant != 0 && variant != 4 && variant != 5)) return false;
    const int64_t tiles = mf16_tiles(d);
    const int64_t M_rows = (int64_t)d->N * d->Ho * d->Wo;
    if (tiles < mf16_min_tiles() || tiles > 0x7fffffff) return false;
    const dim3 grid((unsigned)tiles), block(256);
#define RN_MF16_LAUNCH(G, R, T, S) hipLaunchKernelGGL((conv_igemm_mf16_kernel<G, R, T, S>), grid, block, 0, s, *d, x, w, y, scale, shift, add, mask, add2)
    if (d->w_format == 3) {
        // staged activations (row-coalesced direct-to-LDS loads) where the K loop is long enough to pay for the longer prologue: measured
        // -4 % on the Winograd GEMMs (K = 256) and K >= 1024, +8 % on K = 64 / 128 (profiles/r05_mf16_bounds.txt, 4)
        static const int stg_min_k = [] { const char *e = getenv("RN_MF16_STG_MIN_K"); return e ? atoi(e) : 256; }();
        const bool stg = RN_MF16_STG != 0 && d->kh * d->kw * d->Cin >= stg_min_k;
        // eight waves on a 256 x 128 tile (staged form only; 70 KB of LDS, two workgroups per CU = sixteen waves instead of twelve): the
        // weights' planes are fetched once per 256 rows instead of per 128 -- 0.75 x the bytes through the CU's memory pipe per product.
        // Measured (tools/bench_conv.py, profiles/r05_mf16_bounds.txt, 7): 1x1 1024->256 0.136 -> 0.122 ms, its data gradient 0.129 -> 0.115,
        // 3x3 256->256 -4 %; the per-position launches of the Winograd stage (big -3 %, small +5 %; in the step 29.13 with them on it against
        // 29.0 ms without) keep four waves.  RN_MF16_WV8_MIN: fewest 256-row tiles a launch needs (default 1; 0: never).
        static const int wv8_min = [] { const char *e = getenv("RN_MF16_WV8_MIN"); return e ? atoi(e) : 1; }();
        const int64_t tiles8 = ((M_rows + 255) / 256) * ((d->Cout + 127) / 128);
        if (stg && wv8_min > 0 && d->w_batch_stride == 0 && tiles8 >= wv8_min && tiles8 <= 0x7fffffff) {
            const dim3 grid8((unsigned)tiles8), block8(512);
#define RN_MF16_LAUNCH8(G, R) hipLaunchKernelGGL((conv_igemm_mf16_kernel<G, R, 2, true, 8>), grid8, block8, 0, s, *d, x, w, y, scale, shift, add, mask, add2)
            if (variant == 0) RN_MF16_LAUNCH8(false, true);
            else if (variant == 4) RN_MF16_LAUNCH8(false, false);
            else RN_MF16_LAUNCH8(true, false);
#undef RN_MF16_LAUNCH8
        } else
        if (variant == 0) { if (stg) RN_MF16_LAUNCH(false, true, 2, true); else RN_MF16_LAUNCH(false, true, 2, false); }
        else if (variant == 4) { if (stg) RN_MF16_LAUNCH(false, false, 2, true); else RN_MF16_LAUNCH(false, false, 2, false); }
        else { if (stg) RN_MF16_LAUNCH(true, false, 2, true); else RN_MF16_LAUNCH(true, false, 2, false); }
    } else {
        if (variant == 0) RN_MF16_LAUNCH(false, true, 3, false);
        else if (variant == 4) RN_MF16_LAUNCH(false, false, 3, false);
        else RN_MF16_LAUNCH(true, false, 3, false);
    }
#undef RN_MF16_LAUNCH
    const hipError_t e = hipGetLastError();
    *rc = e == hipSuccess ? RN_OK : (int)e;
    return true;
}

bool rn_igemm_mf16_grouped_launch(const rn_conv_group *g, const float *w, const float *scale, const float *shift, hipStream_t s, int *rc) {
    for (int i = 0; i < g->n; ++i)
        if (!mf16_ok(&g->d[i]) || g->d[i].w_format != g->d[0].w_format) return false;
    rn_conv_group gb = *g;                                  // the caller's tile table counts 128 x 128 tiles: recount
    int64_t total = 0;
    for (int i = 0; i < g->n; ++i) {
        total += mf16_tiles(&g->d[i]);
        gb.tile_end[i] = (int)total;
    }
    for (int i = g->n; i < RN_MAX_GROUP; ++i) gb.tile_end[i] = (int)total;
    if (total < mf16_min_tiles() || total > 0x7fffffff) return false;
    if (g->d[0].w_format == 3) hipLaunchKernelGGL(conv_igemm_mf16_grouped_kernel<2>, dim3((unsigned)total), dim3(256), 0, s, gb, w, scale, shift);
    else hipLaunchKernelGGL(conv_igemm_mf16_grouped_kernel<3>, dim3((unsigned)total), dim3(256), 0, s, gb, w, scale, shift);
    const hipError_t e = hipGetLastError();
    *rc = e == hipSuccess ? RN_OK : (int)e;
    return true;
}
