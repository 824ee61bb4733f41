// bf16 implicit-GEMM convolution, eight-wave 256 x 256 x 64 tile with a phased K loop (round 4) -- the stride-1, same-size layers of
// the bf16 engine (every 3x3 / 1x1 convolution of the heads, the pyramid and the bottlenecks whose input and output planes coincide,
// D/model.py:59-205, and their data gradients, which are convolutions of the same kind), where conv_bf16.hip's tiles -- one barrier per
// 64-byte K-step, whatever their shape -- sit at 750-850 TFLOP/s (profiles/r04_bf16_tile_variants.txt).  Same rn_conv_desc, same packed
// weights, same epilogue arithmetic as conv_bf16.hip; the launchers there choose between the two (bf16_p8_pick).
//
// Structure (tools/probes/gemm8_probe.hip measured it: 1 170-1 240 TFLOP/s on the 8 x 135 x 240, 256 -> 256, 3x3 shape):
//   * 2 x 4 waves, each 128 pixels x 64 channels on v_mfma_f32_16x16x32_bf16 (8 x 4 accumulators of 16 x 16); one workgroup per CU,
//     two K-tiles of both operands in LDS (2 x 64 KB);
//   * an operand tile is [256 rows][128 bytes = 64 k], 16-byte chunk c of row r in slot c ^ ((r >> 1) & 7); it arrives by direct-to-LDS
//     DMA, 8 rows per wave instruction, 8 instructions per wave and K-tile (4 pixels' rows, 4 weights' rows);
//   * a K-tile is four phases of 16 MFMAs (quadrants of the wave's 128 x 64); the fragments of a phase are read from LDS during the
//     phase before, the DMA of the K-tile after next is spread over the phases (3 + 3 + 2 instructions), and there is ONE
//     s_waitcnt vmcnt(0) + s_barrier per K-tile -- 64 MFMAs = ~1000 matrix-core cycles per wave between barriers, against 8 in the old tile;
//   * the padding is a per-lane validity bit per filter row / column (4 + 4 bits for each of the lane's four pixel rows, one VGPR):
//     the pixel rows of a stride-1 same-size layer are one flat sequence, so a tap is a wave-uniform shift of the lane's one byte offset
//     and a lane whose tap falls outside the image sends the out-of-range offset (zero-fill, no memory access);
//   * epilogue straight from the accumulators: v_permlane16_swap pairs the two 16-channel blocks of a lane's row so that a lane holds
//     eight consecutive channels -- 16-byte loads of the addend / mask and 16-byte stores, as in conv_bf16.hip.
// Roofline: MFMA (2.5 PFLOP/s dense bf16).  Algorithmic bytes per tile: (256 + 256) rows x 128 B per K-tile from L2, 128 KB stored.
#include <type_traits>

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int P8_BM = 256, P8_BN = 256, P8_BK = 64, P8_ROWB = 128;
constexpr int P8_OPB = 256 * P8_ROWB;            // bytes of one operand tile: 32 KB
constexpr int P8_BUFB = 2 * P8_OPB;              // one K-tile: pixels' rows, then weights' rows
constexpr int P8_LDS = 2 * P8_BUFB;              // 128 KB (dynamic)
__device__ __forceinline__ int p8_swz(int r) { return (r >> 1) & 7; }

#ifndef P8_STAGGER
#define P8_STAGGER 0                             // experiments (profiles/r04_bf16_p8_stagger.txt): 1 = the wr = 1 waves sleep after each barrier, 2 = they issue MFMAs first
#endif
#ifndef P8_ABL
#define P8_ABL 0                                 // knock-outs for profiles/ (bits): 1 no epilogue, 2 no validity test, 4 no staging after the prologue
#endif
struct P8Tap { int r, s, c; };                   // filter row, filter column, first channel of a K-tile (wave-uniform)

// ---- epilogue: v = scale[c] * acc + shift[c]; [mask before the add]; v += add; [ReLU]; [mask after]; one rounding on the store
// (conv_bf16.hip's arithmetic, in its order).  acc[rb][cb][e] is pixel row rb * 16 + lr, channel cb * 16 + 4 lg + e of the wave's
// 128 x 64 (lr = lane & 15, lg = lane >> 4).  v_permlane16_swap(E, O) exchanges the odd 16-lane rows of E with the even rows of O: with
// E / O the blocks 2 pr / 2 pr + 1, a lane of an even row (lg = 2k) then holds channels 8k .. 8k + 7 of block 2 pr, a lane of an odd row
// those of block 2 pr + 1 -- 16-byte loads of addend / mask and 16-byte stores.  MM: mask mode (0 none, 1 before the add, 2 after the
// activation); SIGN: the result's sign bits are written too (a launch with both a mask and sign_out is not this kernel's: rn_bf16_p8_legal).
template <int MM, bool RELU, bool SIGN>
__device__ __forceinline__ void p8_epilogue(const f32x4 (&acc)[8][4], const rn_conv_desc &d, __bf16 *__restrict__ y, const float *__restrict__ scale,
                                            const float *__restrict__ shift, const __bf16 *__restrict__ add, const __bf16 *__restrict__ mask,
                                            const int mw, const int nw, const int M, const int lane) {
    const int lr = lane & 15, lg = lane >> 4;
    const int half = lg & 1, k8 = lg >> 1;
    int col[2];
    bool col_ok[2];
    float sc[2][8], sh[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        col[pr] = nw + (2 * pr + half) * 16 + 8 * k8;
        col_ok[pr] = col[pr] < d.Cout;                           // Cout % 8 == 0 (launcher): a chunk is inside or outside
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 s4 = (col_ok[pr] && scale != nullptr) ? *reinterpret_cast<const float4 *>(scale + col[pr] + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 h4 = (col_ok[pr] && shift != nullptr) ? *reinterpret_cast<const float4 *>(shift + col[pr] + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            sc[pr][4 * q] = s4.x; sc[pr][4 * q + 1] = s4.y; sc[pr][4 * q + 2] = s4.z; sc[pr][4 * q + 3] = s4.w;
            sh[pr][4 * q] = h4.x; sh[pr][4 * q + 1] = h4.y; sh[pr][4 * q + 2] = h4.z; sh[pr][4 * q + 3] = h4.w;
        }
        if (!col_ok[pr]) col[pr] = 0;                            // loads stay inside the tensors; nothing is stored
    }
    const bool mbits = (d.mask_mode & RN_MASK_BITS) != 0;
    const bool has_add = d.add_mode == 1;
    // A residual layer without a mask (the bottleneck's conv3 + identity + ReLU, D/utils.py:60-80): all sixteen addend chunks of the lane
    // are requested before the first is used -- 64 registers the K loop no longer needs; requested pair by pair they were sixteen
    // serialized round trips per tile (conv_fp8_p8.hip: the same change took its residual 1x1 layers from 2.8 to 3.6 TB/s).
    // With a mask operand (data gradients) both operands are requested for TWO pixel blocks at a time -- four round trips per tile, 32
    // registers -- instead of pair by pair.
    constexpr int RB_AHEAD = MM == 0 ? 8 : 2;                    // pixel blocks whose operands are in flight together
    bf16x8 ads[RB_AHEAD][2], mks[MM == 0 ? 1 : RB_AHEAD][2];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        if (rb % RB_AHEAD == 0) {
#pragma unroll
            for (int r2 = 0; r2 < RB_AHEAD; ++r2) {
                const int m = mw + (rb + r2) * 16 + lr;
                const int64_t rowoff = (int64_t)(m < M ? m : M - 1) * d.Cout;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    if constexpr (MM != 0) {
                        if (!mbits) mks[r2][pr] = *reinterpret_cast<const bf16x8 *>(mask + rowoff + col[pr]);
                    }
                    if (has_add) ads[r2][pr] = *reinterpret_cast<const bf16x8 *>(add + rowoff + col[pr]);
                }
            }
        }
        const int m = mw + rb * 16 + lr;
        const bool row_ok = m < M;
        const int64_t rowoff = (int64_t)(row_ok ? m : M - 1) * d.Cout;
        unsigned mk[2];                                          // MM != 0: bit j = the mask of channel j of the pair's chunk
        bf16x8 ad[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int64_t off = rowoff + col[pr];
            if constexpr (MM != 0) {
                if (mbits) {
                    mk[pr] = rn_sign_bits(mask, off, 8);
                } else {
                    const bf16x8 t = mks[rb % RB_AHEAD][pr];
                    unsigned b = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) b |= (unsigned)((float)t[j] > 0.f) << j;
                    mk[pr] = b;
                }
            }
            const bf16x8 zero = {};
            ad[pr] = has_add ? ads[rb % RB_AHEAD][pr] : zero;
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[rb][2 * pr][e]), __float_as_uint(acc[rb][2 * pr + 1][e]), false, false);
                v[e] = __uint_as_float(r[0]);
                v[4 + e] = __uint_as_float(r[1]);
            }
            bf16x8 o;
            unsigned sb = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float u = v[j] * sc[pr][j] + sh[pr][j];
                if constexpr (MM == 1) u = (mk[pr] >> j) & 1u ? u : 0.f;
                u += (float)ad[pr][j];                           // no addend: + 0
                if constexpr (RELU) u = fmaxf(u, 0.f);
                if constexpr (MM == 2) u = (mk[pr] >> j) & 1u ? u : 0.f;
                o[j] = (__bf16)u;
                if constexpr (SIGN) sb |= (unsigned)((float)o[j] > 0.f) << j;   // the sign of what is STORED
            }
            const bool ok = row_ok && col_ok[pr];
            if (ok) *reinterpret_cast<bf16x8 *>(y + rowoff + col[pr]) = o;
            if constexpr (SIGN) {
                // the word of channels 32 pr .. 32 pr + 31 of the wave's 64: this lane's byte sits at 2 half + k8; the four lanes of a
                // pixel are 16 apart.  Cout % 32 == 0: the four chunks of a word are inside or outside together.
                unsigned wd = sb << (8 * (2 * half + k8));
                wd |= (unsigned)__shfl_xor((int)wd, 16, 64);
                wd |= (unsigned)__shfl_xor((int)wd, 32, 64);
                if (ok && lg == 0) reinterpret_cast<unsigned *>(d.sign_out)[(rowoff + col[pr]) >> 5] = wd;
            }
        }
    }
}

__device__ __forceinline__ void p8_tile(const rn_conv_desc &d, const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                        __bf16 *__restrict__ y, const float *__restrict__ scale, const float *__restrict__ shift,
                                        const __bf16 *__restrict__ add, const __bf16 *__restrict__ mask, const int tile, char *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave >> 2, wc = wave & 3;     // rows wr * 128, columns wc * 64 of the tile
    const int ntn = (d.Cout + P8_BN - 1) / P8_BN;
    const int m0 = (tile / ntn) * P8_BM, n0 = (tile % ntn) * P8_BN;
    const int HoWo = d.Ho * d.Wo;
    const int M = d.N * HoWo;                    // < 2^31 - 256 (launcher)
    const int Cin = d.Cin, K = d.kh * d.kw * Cin, nkt = K / P8_BK;

    // descriptors.  Pixels: the flat [M][Cin] tensor from `halo` rows in front of the tile (the farthest a tap reaches back).
    const int ab = d.b < 0 ? -d.b : d.b;
    const int halo = ((d.p < 0 ? -d.p : d.p) + (d.kh - 1) * ab) * d.Wi + (d.p_w < 0 ? -d.p_w : d.p_w) + (d.kw - 1) * ab;
    const int base_row = m0 > halo ? m0 - halo : 0;
    const int64_t a_bytes = ((int64_t)M - base_row) * Cin * 2;
    const v4i32 rs_a = make_rsrc(x + (int64_t)base_row * Cin, (unsigned)(a_bytes > 0x7FFFFFFF ? 0x7FFFFFFF : a_bytes));
    const v4i32 rs_b = make_rsrc(w, (unsigned)((int64_t)d.Cout * K * 2));
    const unsigned lds0 = lds_addr(lds);

    // ---- staging.  Instruction i of this wave fills rows 64 (i & 3) + 8 wave + (lane >> 3) of the pixels (i < 4) or the weights: the
    // swizzle of the row does not depend on i, so ONE lane offset per operand and the i-th instruction adds 64 rows on the scalar side.
    const int row0 = 8 * wave + (lane >> 3);
    const int chunk = (lane & 7) ^ p8_swz(row0);
    const unsigned voff_a = (unsigned)((m0 - base_row + row0) * Cin * 2 + chunk * 16);
    const unsigned voff_b = (unsigned)((n0 + row0) * K * 2 + chunk * 16);   // rows past Cout: past the descriptor's range
    // validity of the lane's four pixel rows: bits 8 j + r = filter row r reads inside the image, bits 8 j + 4 + s = filter column s
    unsigned pk = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + row0 + 64 * j;
        if (m < M) {
            const unsigned rem = (unsigned)m % (unsigned)HoWo;
            const int oh = (int)(rem / (unsigned)d.Wo), ow = (int)(rem - (unsigned)oh * (unsigned)d.Wo);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < d.kh && (unsigned)(oh + d.p + t * d.b) < (unsigned)d.Hi) pk |= 1u << (8 * j + t);
                if (t < d.kw && (unsigned)(ow + d.p_w + t * d.b) < (unsigned)d.Wi) pk |= 1u << (8 * j + 4 + t);
            }
        }
    }
    auto uni = [](const v4i32 r) {
        v4i32 o;
        o.x = __builtin_amdgcn_readfirstlane(r.x); o.y = __builtin_amdgcn_readfirstlane(r.y);
        o.z = __builtin_amdgcn_readfirstlane(r.z); o.w = __builtin_amdgcn_readfirstlane(r.w);
        return o;
    };
    // The pixel offsets of the four staging instructions AT THE CURRENT TAP (validity applied): recomputed when the staging cursor enters
    // a new tap (every Cin / 64 K-tiles), so that a K-tile's staging is four buffer loads with a scalar channel offset and no vector
    // arithmetic.  (Counters, profiles/r04_pmc_p8.txt: 2.1 VALU instructions per MFMA -- with the MFMA's own 8 issue cycles that fills
    // the 16-cycle slot; the per-instruction offset arithmetic was a fifth of them.)
    unsigned va[4];
    auto set_tap = [&](const P8Tap &tp) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sh = __builtin_amdgcn_readfirstlane((((d.p + tp.r * d.b) * d.Wi + d.p_w + tp.s * d.b + 64 * i) * Cin) * 2);
            const unsigned ok = (P8_ABL & 2) ? 1u : (pk >> (8 * i + (tp.r & 3))) & (pk >> (8 * i + 4 + (tp.s & 3))) & 1u;
            va[i] = ok ? voff_a + (unsigned)sh : 0x80000000u;
        }
    };
    auto dma = [&](const int i, const P8Tap &tp, const int kt, const int buf) {   // instruction i of this wave for K-tile kt (at tap tp: set_tap)
        const unsigned dst = lds0 + (unsigned)(buf * P8_BUFB + (wave_u + 8 * i) * 1024);
        if (i < 4) {
            dma16(uni(rs_a), dst, va[i], (unsigned)__builtin_amdgcn_readfirstlane(tp.c * 2));
        } else {
            dma16(uni(rs_b), dst, voff_b, (unsigned)__builtin_amdgcn_readfirstlane((kt * P8_BK + (i - 4) * 64 * K) * 2));
        }
    };
    auto next_tap = [&](P8Tap &tp) {
        tp.c += P8_BK;
        if (tp.c == Cin) { tp.c = 0; if (++tp.s == d.kw) { tp.s = 0; ++tp.r; } }
    };

    // ---- fragments: block rb (16 rows) of this wave's 128 pixel rows, k half kh; block cb of its 64 weight rows.  One address per
    // operand and lane: the block index adds 16 rows = 2048 bytes (16 rows do not change (r >> 1) & 7), the k half flips chunk bit 2.
    const int lr = lane & 15, lg = lane >> 4;
    const int ra = wr * 128 + lr, rbb = wc * 64 + lr;
    const int a0 = ra * P8_ROWB + 16 * (lg ^ p8_swz(ra));
    const int b0 = P8_OPB + rbb * P8_ROWB + 16 * (lg ^ p8_swz(rbb));
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa0[4][2], fa1[4][2], fb0[2][2], fb1[2][2];           // pixel quadrant 0 / 1 (64 rows each), weight quadrant 0 / 1 (32 rows each)
    auto rd = [&](const char *S, int off) { return *reinterpret_cast<const bf16x8 *>(S + off); };
    auto read_a = [&](bf16x8 (&f)[4][2], const char *S, int qm) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) f[i][kh] = rd(S, (a0 ^ (64 * kh)) + (4 * qm + i) * 2048);
    };
    auto read_b = [&](bf16x8 (&f)[2][2], const char *S, int qn) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) f[j][kh] = rd(S, (b0 ^ (64 * kh)) + (2 * qn + j) * 2048);
    };
    // weight fragment first: a lane then holds 4 consecutive channels of one pixel
    auto mma = [&](const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2], int qm, int qn) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * qm + i][2 * qn + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kh], fa[i][kh], acc[4 * qm + i][2 * qn + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: K-tile 0 -> buffer 0, K-tile 1 -> buffer 1 (all 8 instructions each); wait for tile 0, read its first fragments
    P8Tap t1 = {0, 0, 0};
    set_tap(t1);
#pragma unroll
    for (int i = 0; i < 8; ++i) dma(i, t1, 0, 0);
    next_tap(t1);
    if (nkt > 1) {
        if (t1.c == 0) set_tap(t1);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(i, t1, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    P8Tap t2 = t1;                                              // the tap of K-tile t + 2 (in body(t))
    next_tap(t2);
    if (t2.c == 0 && nkt > 2) set_tap(t2);                      // va[] always belongs to t2's tap from here on
    asm volatile("s_barrier" ::: "memory");
    read_a(fa0, lds, 0);
    read_b(fb0, lds, 0);

    // ---- main loop: K-tile t in buffer t & 1.  The weight fragment sets swap roles every K-tile (x = the set whose fragments are already
    // there, y = the other): the order of the four quadrants is (0,x) (0,y) (1,y) (1,x), and the last phase -- which still needs x and
    // pixel set 1 -- loads the NEXT tile's pixel set 0 and its weight quadrant qy into the y set: the next tile starts from (0, y).
    // Before the barrier of tile t every fragment of the tile is in registers and tile t + 1 has been requested completely (its last
    // instruction a phase ago); after it buffer t & 1 is free for tile t + 2 and buffer (t + 1) & 1 is readable.
    auto body = [&](auto late_tag, const int t, bf16x8 (&fbx)[2][2], bf16x8 (&fby)[2][2], const int qx, const int qy) {
        constexpr bool LATE = decltype(late_tag)::value;        // P8_STAGGER 2: the wr = 1 waves issue a phase's MFMAs BEFORE its reads / staging
        const char *S = lds + (t & 1) * P8_BUFB;
        const char *Sn = lds + ((t + 1) & 1) * P8_BUFB;
        const bool more = t + 1 < nkt, stage = t + 2 < nkt && !(P8_ABL & 4);
        if (!LATE) read_b(fby, S, qy);
        mma(fa0, fbx, 0, qx);
        if (LATE) { __builtin_amdgcn_sched_barrier(0); read_b(fby, S, qy); }
        if (!LATE) read_a(fa1, S, 1);
        mma(fa0, fby, 0, qy);
        if (LATE) { __builtin_amdgcn_sched_barrier(0); read_a(fa1, S, 1); }
        mma(fa1, fby, 1, qy);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (P8_STAGGER == 1 && wr) __builtin_amdgcn_s_sleep(2);
        if (!LATE && more) {
            read_a(fa0, Sn, 0);
            read_b(fby, Sn, qy);
        }
        // the last quadrant, with the WHOLE staging of K-tile t + 2 (its buffer was released by the barrier) between its MFMAs: one
        // instruction per two MFMAs, so that every load has the three phases of K-tile t + 1 to land in (spread over phases 0, 1 and 3 as
        // at first, the last ones had a single phase: -24 % with the staging knocked out, profiles/r04_bf16_p8_knockouts.txt)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int kh = g >> 2, i = g & 3;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[4 + i][2 * qx + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbx[j][kh], fa1[i][kh], acc[4 + i][2 * qx + j], 0, 0, 0);
            if (!LATE && stage) dma(g, t2, t + 2, t & 1);
        }
        __builtin_amdgcn_s_setprio(0);
        if (LATE) {
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                read_a(fa0, Sn, 0);
                read_b(fby, Sn, qy);
            }
            if (stage) {
#pragma unroll
                for (int g = 0; g < 8; ++g) dma(g, t2, t + 2, t & 1);
            }
        }
        next_tap(t2);
        if (t2.c == 0 && t + 3 < nkt) set_tap(t2);              // the staging cursor entered a new tap
    };
    if (P8_STAGGER == 2 && wr) {
        for (int t = 0; t < nkt; t += 2) {
            body(std::true_type{}, t, fb0, fb1, 0, 1);
            if (t + 1 < nkt) body(std::true_type{}, t + 1, fb1, fb0, 1, 0);
        }
    } else {
        for (int t = 0; t < nkt; t += 2) {
            body(std::false_type{}, t, fb0, fb1, 0, 1);
            if (t + 1 < nkt) body(std::false_type{}, t + 1, fb1, fb0, 1, 0);
        }
    }

    // ---- epilogue (p8_epilogue)
    if (P8_ABL & 1) {                                          // keep the accumulators alive, store next to nothing
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][3];
        if (t == 1.2345e-30f) y[0] = (__bf16)t;
        return;
    }
    // one straight-line instance per epilogue form (the flags are wave-uniform: as run-time tests inside the 128 elements of a lane they
    // compiled to ~10 000 lines of branches, 14 % of the kernel's time)
    const int mmode = d.mask_mode & 3;
    const bool relu = d.act == 1, sign = d.sign_out != nullptr;
#define P8_EPI(MM, RELU, SIGN) p8_epilogue<MM, RELU, SIGN>(acc, d, y, scale, shift, add, mask, m0 + wr * 128, n0 + wc * 64, M, lane)
    if (mmode == 0) {
        if (sign) { if (relu) P8_EPI(0, true, true); else P8_EPI(0, false, true); }
        else { if (relu) P8_EPI(0, true, false); else P8_EPI(0, false, false); }
    } else if (mmode == 1) {
        if (relu) P8_EPI(1, true, false); else P8_EPI(1, false, false);
    } else {
        if (relu) P8_EPI(2, true, false); else P8_EPI(2, false, false);
    }
#undef P8_EPI
}

__global__ __launch_bounds__(512, 2) void conv_igemm_bf16_p8_kernel(const rn_conv_desc d, const __bf16 *__restrict__ x, const __bf16 *__restrict__ w,
                                                                   __bf16 *__restrict__ y, const float *__restrict__ scale,
                                                                   const float *__restrict__ shift, const __bf16 *__restrict__ add,
                                                                   const __bf16 *__restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) char p8_lds[];
    p8_tile(d, x, w, y, scale, shift, add, mask, xcd_remap(blockIdx.x, gridDim.x), p8_lds);
}

// Grouped launch (rn_conv_igemm_bf16_grouped): the pyramid levels of a head layer as ONE grid; a workgroup finds its problem by tile id.
__global__ __launch_bounds__(512, 2) void conv_igemm_bf16_p8_grouped_kernel(const rn_conv_group g, const __bf16 *__restrict__ w,
                                                                           const float *__restrict__ scale, const float *__restrict__ shift) {
    extern __shared__ __attribute__((aligned(16))) char p8_lds[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int i = 0; i < RN_MAX_GROUP - 1; ++i) p += (i + 1 < g.n && tile >= g.tile_end[i]) ? 1 : 0;
    p = __builtin_amdgcn_readfirstlane(p);
    const int first = p > 0 ? g.tile_end[p - 1] : 0;
    p8_tile(g.d[p], reinterpret_cast<const __bf16 *>(g.x[p]), w, reinterpret_cast<__bf16 *>(g.y[p]), scale, shift,
            reinterpret_cast<const __bf16 *>(g.add[p]), reinterpret_cast<const __bf16 *>(g.mask[p]), tile - first, p8_lds);
}

// ---------------------------------------------------------------------------------------------- host side (used by conv_bf16.hip)
// What the kernel can compute: a stride-1 convolution whose output plane is the input plane (so that pixel rows are one flat sequence
// on both sides), batch-dense NHWC operands, Cin a multiple of the 64-channel K-tile, at most 4 x 4 taps, a dense bf16 result.
bool rn_bf16_p8_legal(const rn_conv_desc *d, int y_is_f32) {
    if (y_is_f32 || d->a != 1 || d->div_shift != 0 || d->Hi != d->Ho || d->Wi != d->Wo) return false;
    if (d->act == 2 || (d->mask_mode != 0 && d->sign_out != nullptr)) return false;      // epilogue forms the kernel has no instance of
    if (d->Cin < 64 || (d->Cin & 63) || (d->Cout & 7) || d->kh > 4 || d->kw > 4) return false;
    const int64_t plane = (int64_t)d->Hi * d->Wi;
    if (d->x_batch_stride != plane * d->Cin || d->y_batch_stride != plane * d->Cout) return false;
    if (d->os != 1 || d->oo_h != 0 || d->oo_w != 0 || d->Hy != d->Ho || d->Wy != d->Wo || d->add_mode == 2) return false;
    if (d->add_mode == 1 && d->add_batch_stride != d->y_batch_stride) return false;
    const int64_t K = (int64_t)d->kh * d->kw * d->Cin, M = (int64_t)d->N * plane;
    const int64_t ab = d->b < 0 ? -d->b : d->b;
    const int64_t halo = (llabs((long long)d->p) + (d->kh - 1) * ab) * d->Wi + llabs((long long)d->p_w) + (d->kw - 1) * ab;
    if (M + 256 > 0x7fffffffLL || (256 + 2 * halo + 64) * d->Cin * 2 > 0x7fffffffLL || ((int64_t)d->Cout + 256) * K * 2 > 0x7fffffffLL) return false;
    return true;
}
// Whether the launchers take it: RN_OPT_BF16_P8 = 0 never, 2 wherever legal, 1 (default) where it has measured faster -- see the rule's
// comment in conv_bf16.hip.
int rn_bf16_p8_launch(const rn_conv_desc *d, const void *x, const void *w, void *y, const float *scale, const float *shift,
                      const void *add, const void *mask, hipStream_t stream) {
    static const hipError_t attr = hipFuncSetAttribute((const void *)conv_igemm_bf16_p8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    if (attr != hipSuccess) return (int)attr;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    const int64_t tiles = ((M + 255) / 256) * ((d->Cout + 255) / 256);
    if (tiles > 0x7fffffff) return RN_EINVAL;
    hipLaunchKernelGGL(conv_igemm_bf16_p8_kernel, dim3((unsigned)tiles), dim3(512), P8_LDS, stream, *d, reinterpret_cast<const __bf16 *>(x),
                       reinterpret_cast<const __bf16 *>(w), reinterpret_cast<__bf16 *>(y), scale, shift, reinterpret_cast<const __bf16 *>(add),
                       reinterpret_cast<const __bf16 *>(mask));
    RN_LAUNCH_CHECK();
    return RN_OK;
}
int rn_bf16_p8_launch_grouped(const rn_conv_group *g, int tiles, const void *w, const float *scale, const float *shift, hipStream_t stream) {
    static const hipError_t attr = hipFuncSetAttribute((const void *)conv_igemm_bf16_p8_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(conv_igemm_bf16_p8_grouped_kernel, dim3((unsigned)tiles), dim3(512), P8_LDS, stream, *g, reinterpret_cast<const __bf16 *>(w), scale, shift);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
