// Weight gradient of the convolutions on the fp32 matrix cores.
//
// The reference obtains dW from torch autograd (cuDNN wgrad kernels) for every nn.Conv2d of D/model.py; here it
// is one GEMM per layer:  dW[co][r][s][ci] = sum over output pixels of dY[pixel][co] * X[pixel shifted by the
// tap][ci], i.e.  M = Cout,  N = kh*kw*Cin (the packed-weight row, so a tile may span several taps of a
// narrow layer),  K = N*Ho*Wo pixels.
//
// Both operands are pixel-major in memory (NHWC: one pixel = one contiguous channel vector), which for THIS product
// is reduction-major -- the layout the MFMA operands want anyway.  So the tiles go global -> LDS exactly as they lie,
// [32 pixels][channels], and they go there directly: buffer loads with the LDS bit set (16 bytes per lane, a wave
// instruction fills 1 KiB of contiguous LDS = 256/BM pixels of a tile), no staging registers, no ds_write, no store
// phase.  Everything that must read as zero -- image padding, pixels past the end of the K slice, columns past the
// matrix -- is given an out-of-range offset and the buffer range check writes 0.0 for it (per dword, soffset included;
// tools/probes/buffer_probe.hip, lds_dma_probe.hip), so there are no clamps, masks or selects either.  Per wave and
// K-step that leaves: 8 DMA instructions, for the X half one pixel-table read + 6 VALU each, for the dY half nothing
// (one offset register fixed for the whole kernel, the pixel advance rides in the scalar soffset).
// Why it matters: timing the kernel with pieces knocked out showed that whatever a wave does outside its MFMA stream is
// paid in full -- it is NOT hidden behind the partner wave's MFMAs -- so the staging work had to go, not move.
//
// Fragments: v_mfma_f32_32x32x2_f32 wants A[row i = lane&31][k = lane>>5] in one register per lane.  For dY a lane reads
// TWO adjacent channels of pixel k with one ds_read_b64 and uses them as row i of two different 32-row MFMA tiles (tile
// tm holds channels 2*i + tm: a fixed permutation of the wave's 64 channels, undone by the epilogue's index arithmetic).
// For X it reads channels j and j + 32 with one ds_read2_b32, so that tile tn holds 32 CONTIGUOUS columns and an atomic
// instruction of the epilogue covers two full 128-byte row segments (the full-rate shape; stride-2 columns touched
// twice the cache lines and cost the short-K 1x1 layers up to 18 %).  Both reads are conflict-free, every fragment
// address is base + immediate; per wave and 16-pixel K-step 16 LDS reads against 32 MFMAs (64 cycles each).
//
// K is split over the grid; partial tiles are added with fp32 atomics into a buffer the caller zeroes -- the five
// pyramid levels of a shared head accumulate into the same buffer.  All tiles of one K-slice run on the same XCD so
// they stream the same dY / X slabs through one L2.
//
// The column sums of dY (bias / batch-norm beta gradients, and the mean term of the gamma gradient) ride along: the
// waves of N-tile 0 that own distinct channels add up their A fragments (two v_add per four MFMAs, inside the MFMA
// shadow) and finish with one atomic per channel.
//
// Roofline: MFMA (fp32 157.3 TF).
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "conv_wgrad_geom.h"

// RN_WG_KO (RN_EXPERIMENT builds; timing only, wrong results): 1 = the tile-sized atomic accumulation left out, 2 = workgroup-scope atomics
#ifndef RN_WG_KO
#define RN_WG_KO 0
#endif
#if RN_WG_KO == 1
#define RN_WG_ATOMIC(P, V) do { if ((V) == 1.2345e-30f) atomicAdd((P), (V)); } while (0)
#elif RN_WG_KO == 2
#define RN_WG_ATOMIC(P, V) __hip_atomic_fetch_add((P), (V), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define RN_WG_ATOMIC(P, V) atomicAdd((P), (V))
#endif
#include "mfma_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define WK_MAX 32                // pixels per K-step: 32 or 16 (the kernel's WK parameter)

struct WgradArgs {
    const float *dy, *x;
    float *dw;
    float *colsum;               // [Cout] accumulated sum over pixels of dy, or NULL
    int ldy, N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad, in_relu;
    int Kflat, Kpad;             // kh*kw*Cin and its round-up to 32
    int tiles_n;                 // number of N tiles
    int tiles, splits;           // tiles_m * tiles_n, K slices
    int xcd_map;                 // 1: whole K slices per XCD (see the kernel)
    int tiles_per_batch;         // tiles of ONE problem; tiles = tiles_per_batch * batch count
    int colsum_batch;            // the batch entry whose dy column sums are wanted
    int64_t dy_bstride, x_bstride, dw_bstride;   // floats between the operands of consecutive batch entries
    int64_t pixels, per_split;   // K extent and K per slice (multiple of WK)
    // Deterministic form (RN_OPT_DETERMINISTIC): instead of atomics, K slice s stores its partial tiles into slab s -- an image
    // of the whole result, slab_stride floats -- and its column sums into cs_slab + s * Cout; wgrad_combine_kernel adds the slabs
    // in slice order.  NULL: fp32 atomics.
    float *slab, *cs_slab;
    int64_t slab_stride;
    // RN_FP32_SPLIT3 (fp16 two-term products; mfma_split.h): the amax tables of dy and x and how many images they cover (count > 0), or
    // ONE plain word each (count -1: a Winograd-domain tensor's), else NULL.  The reduction runs over the pixels of ALL images, so each
    // operand takes ONE power-of-two scale: that of the largest exponent in any of its tables.
    const void *dy_amax, *x_amax;
    int dy_amax_n, x_amax_n;
};

// scale / inverse scale of an operand (all 64 lanes of the wave active; the result is wave-uniform)
__device__ __forceinline__ void wgrad_scales(const void *amax, int count, float &scale, float &unscale) {
    const int e = count > 0 ? rn_amax_exp_all(amax, count)
                            : (int)((__builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned *>(amax)) >> 23) & 0xffu);
    const int se = rn_f16_scale_exp_of(e);
    scale = rn_exp_to_float(se);
    unscale = rn_exp_to_float(254 - se);
}

// WK pixels per K-step, OCC workgroups per CU the register budget is cut for (LDS: 2 * WK * (BM + BN) * 4 + 8 KiB).
// SPLIT: split-operand products (mfma_split.h): the 16 pixels of a K-step are one v_mfma_f32_32x32x16_bf16 k extent, a lane's
// eight of them are the ones it reads anyway (pixels 2j + (lane >> 5)), for both operands.
// HALF (with SPLIT): the fp16 two-term form (RN_FP32_SPLIT3): both operands scaled by their tensors' powers of two inside the split,
// three v_mfma_f32_32x32x16_f16 per block, the tile multiplied by the two inverse scales before it leaves.
template <int WM, int WN, bool RELU, int WK, int OCC, bool SPLIT = false, bool HALF = false>
__global__ __launch_bounds__(256, OCC) void conv_wgrad_kernel(const WgradArgs p) {
    static_assert(!SPLIT || WK == 16, "split-operand form: 16-pixel K-steps");
    static_assert(!HALF || SPLIT, "the fp16 form is a split form");
    using G = WgradGeom<WM, WN, WK>;                         // index arithmetic shared with the host-side range check
    constexpr int BM = G::BM, BN = G::BN;
    constexpr int TB = G::TB;                                // K-steps per pixel-table batch: one entry per thread
    constexpr int CA = G::CA, CB = G::CB;                    // 16-byte chunks per tile row
    constexpr int PA = G::PA, PB = G::PB;                    // pixels one wave instruction (64 lanes x 16 B) covers
    constexpr int IA = G::IA, IB = G::IB;                    // DMA instructions per wave per K-step (4 waves share a tile)
    __shared__ float lds[2][WK * (BM + BN)];                 // per buffer: A [32 px][BM], then B [32 px][BN]
    __shared__ int4 pixtab[2][TB * WK];                      // two batches of TB K-steps: (x byte offset, ih0, iw0, -)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroup id -> (tile, K slice).  Ids are dealt round-robin to the 8 XCDs; all tiles of one K slice read the same
    // dY / X slabs, so a slice is kept on ONE XCD (slice s on XCD s % 8, its tiles consecutive there) and the slabs
    // stream through that L2 once instead of through all eight.
    // Measured: a win (+4..10 %) when a slice has many tiles to share the slabs (>= 16: the 256-channel 3x3 layers), a
    // loss for the few-tile shapes, which keep the plain order (consecutive ids = the tiles of one slice).
    int slice, tile;
    if (p.xcd_map) {
        const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
        slice = (wi / p.tiles) * 8 + xcd;
        tile = wi % p.tiles;
    } else {
        slice = blockIdx.x / p.tiles;
        tile = blockIdx.x % p.tiles;
    }
    if (slice >= p.splits) return;                           // padding of the last group of 8 slices
    // Batched form (the 36 positions of the Winograd weight gradient): independent problems of identical shape, operands
    // and result p.*_bstride apart; a tile index enumerates (batch entry, tile).
    const int batch = tile / p.tiles_per_batch;
    tile -= batch * p.tiles_per_batch;
    const float *dy_b = p.dy + (int64_t)batch * p.dy_bstride;
    const float *x_b = p.x + (int64_t)batch * p.x_bstride;
    float *dw_b = p.dw + (int64_t)batch * p.dw_bstride;
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int64_t kbeg = (int64_t)slice * p.per_split;
    const int64_t kend = (kbeg + p.per_split < p.pixels) ? kbeg + p.per_split : p.pixels;
    const int nks = (int)((kend - kbeg + WK - 1) / WK);
    if (nks <= 0) return;
    const int HoWo = p.Ho * p.Wo;

    // Buffer descriptors (scalar registers).
    //   dY: the K slice [kbeg, kend) only, so "pixel >= kend" is out of range by itself;
    //   X : from the first image the slice touches; the host guarantees that the span of images of one slice stays
    //       below 2 GiB, so byte offsets are plain int32 and -1 is always out of range.
    const int n_first = (int)(kbeg / HoWo);
    const int rel0 = (int)(kbeg - (int64_t)n_first * HoWo);  // index of pixel kbeg within image n_first
    const int64_t img = (int64_t)p.Hi * p.Wi * p.Cin;        // floats per input image
    int64_t xbytes = ((int64_t)p.N - n_first) * img * 4;
    if (xbytes > 0x7FFFFFFF) xbytes = 0x7FFFFFFF;
    const v4i32 rs_a = make_rsrc(dy_b + kbeg * p.ldy, (unsigned)((kend - kbeg) * p.ldy * 4));
    const v4i32 rs_b = make_rsrc(x_b + (int64_t)n_first * img, (unsigned)xbytes);

    // Lane -> (pixel within the instruction's group, 16-byte chunk).  Instruction j of wave w fills pixels
    // (w * I + j) * P + lane / C of the tile: 1 KiB of LDS starting at that pixel's row.
    const int ca = lane % CA, pa = lane / CA;
    const int cb = lane % CB;                                // (the B half's pixel is G::tab_index's business)
    const bool a_col_ok = (m0 + 4 * ca) < p.ldy;
    const unsigned a_voff = a_col_ok ? (unsigned)(((wave * IA * PA + pa) * p.ldy + m0 + 4 * ca) * 4) : 0x80000000u;
    const unsigned a_step = (unsigned)(PA * 4) * (unsigned)p.ldy;          // bytes between consecutive instructions (scalar)
    const int jcol = n0 + 4 * cb;                            // B: the chunk fixes (tap, first input channel)
    const int tap = jcol / p.Cin;
    const int ci0 = jcol - tap * p.Cin;
    const int fr = tap / p.kw, fs = tap - fr * p.kw;
    const int fr_t = jcol < p.Kflat ? fr : (1 << 24);        // a column past the matrix fails every row test below
    const int tap_off = ((fr * p.Wi + fs) * p.Cin + ci0) * 4;

    // Pixel table: one entry per pixel of the K range, (byte offset of input pixel (ih0, iw0) channel 0, ih0, iw0) with
    // ih0 = oh*stride - pad: the two integer divisions of the (n, oh, ow) decomposition are done once per pixel for all
    // lanes and all taps.  Filled TB K-steps at a time by the whole workgroup (one entry per thread).
    auto fill_batch = [&](int j) {
        const int rel = rel0 + j * (TB * WK) + tid;
        int4 e = make_int4(0, -(1 << 28), 0, 0);              // past the slice: fails the row test (also with fr_t added)
        if (kbeg + (int64_t)j * (TB * WK) + tid < kend) {
            const unsigned n = (unsigned)rel / (unsigned)HoWo;
            const unsigned rem = (unsigned)rel - n * (unsigned)HoWo;
            const unsigned oh = rem / (unsigned)p.Wo, ow = rem - oh * (unsigned)p.Wo;
            const int ih0 = (int)oh * p.stride - p.pad, iw0 = (int)ow * p.stride - p.pad;
            e = make_int4((((int)n * p.Hi + ih0) * p.Wi + iw0) * p.Cin * 4, ih0, iw0, 0);
        }
        pixtab[j & 1][tid] = e;
    };
    // The X half's offsets are software-pipelined one K-step ahead of their loads: the table entries of step ks+2 are
    // read right after the loads of step ks+1 went out, and turned into offsets half-way through the MFMA stream of step
    // ks (inside its shadow).  So the load phase at the top of an iteration is eight instructions with ready operands --
    // no LDS round trip, no VALU.  (Measured on the 256-channel 3x3 layers: the read -> test -> load chain at the top
    // cost 4-5 %, the table fill 2.5 %, unconditional column-sum adds 3 %.)
    unsigned b_voff[IB];                                     // offsets of the next step to issue
    int4 e[IB];                                              // table entries of the step after that
    auto table_read = [&](int ks) {
        const int4 *tab = pixtab[(ks / TB) & 1];
#pragma unroll
        for (int j = 0; j < IB; ++j) e[j] = tab[G::tab_index(ks, wave, lane, j)];
    };
    auto make_offsets = [&]() {
#pragma unroll
        for (int j = 0; j < IB; ++j) {
            const bool ok = ((unsigned)(e[j].y + fr_t) < (unsigned)p.Hi) & ((unsigned)(e[j].z + fs) < (unsigned)p.Wi);
            b_voff[j] = ok ? (unsigned)(e[j].x + tap_off) : 0xFFFFFFFFu;
        }
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = lds_addr(&lds[0][0]);
    // Issue the loads of K-step ks into buffer `buf` (b_voff holds that step's offsets).  The caller's barrier has made
    // sure nobody still reads the buffer.
    auto dma_step = [&](int ks, int buf) {
        const unsigned A = lds0 + (unsigned)((buf * G::BUF + G::dma_a(wave_u, 0)) * 4);   // this wave's first row
        const unsigned B = lds0 + (unsigned)((buf * G::BUF + G::dma_b(wave_u, 0)) * 4);
        const unsigned so = (unsigned)ks * (unsigned)(WK * 4) * (unsigned)p.ldy;   // scalar: the K-step advance
#pragma unroll
        for (int j = 0; j < IA; ++j) dma16(rs_a, A + j * (PA * BM * 4), a_voff, so + j * a_step);
#pragma unroll
        for (int j = 0; j < IB; ++j) dma16(rs_b, B + j * (PB * BN * 4), b_voff[j], 0u);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e_ = 0; e_ < 16; ++e_) acc[i][j][e_] = 0.f;
    float2 cs = make_float2(0.f, 0.f);                       // column sums of this lane's two channels (its k parity)
    float s_a = 1.f, s_b = 1.f, us = 1.f;                    // HALF: operand scales and the product of their inverses
    if constexpr (HALF) {
        float ua, ub;
        wgrad_scales(p.dy_amax, p.dy_amax_n, s_a, ua);
        wgrad_scales(p.x_amax, p.x_amax_n, s_b, ub);
        us = ua * ub;
    }

    fill_batch(0);
    __syncthreads();
    table_read(0);
    make_offsets();
    dma_step(0, 0);
    table_read(nks > 1 ? 1 : 0);
    make_offsets();                                          // offsets of step 1
    rn_wait_dma();
    __syncthreads();

    const int fa0 = G::frag_a(wm, lane, 0);
    const int fb0 = G::frag_b(wn, lane, 0) - WK * BM;         // B: channels j and j + 32 (see the epilogue)
    // Only the waves that own distinct channels of N-tile 0 add up column sums: two copies of the loop, chosen once.
    const bool do_cs = p.colsum != nullptr && batch == p.colsum_batch && (tile % p.tiles_n) == 0 && wn == 0;
    auto k_loop = [&](auto cs_tag) {
        constexpr bool CS = decltype(cs_tag)::value;
        for (int ks = 0; ks < nks; ++ks) {
            const int buf = ks & 1;
            if (ks + 1 < nks) dma_step(ks + 1, buf ^ 1);
            table_read(ks + 2 < nks ? ks + 2 : nks - 1);      // its batch was published by an earlier barrier (see below)
            const float *A = lds[buf] + fa0;
            const float *B = lds[buf] + WK * BM + fb0;
            if constexpr (SPLIT) {
                float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
                for (int kp = 0; kp < 8; ++kp) {
                    const float2 va = *reinterpret_cast<const float2 *>(A + 2 * kp * BM);
                    float2 vb = make_float2(B[2 * kp * BN], B[2 * kp * BN + 32]);
                    if (RELU) { vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); }
                    if (CS) { cs.x += va.x; cs.y += va.y; }
                    a0[kp] = va.x; a1[kp] = va.y; b0[kp] = vb.x; b1[kp] = vb.y;
                }
                if constexpr (HALF) {
                    const SplitH8 sa0 = split8h(a0, s_a), sb0 = split8h(b0, s_b), sb1 = split8h(b1, s_b);
                    RN_SPLITH_MFMA(acc[0][0], sa0, sb0);
                    const SplitH8 sa1 = split8h(a1, s_a);
                    RN_SPLITH_MFMA(acc[0][1], sa0, sb1);
                    make_offsets();
                    if ((ks % TB) == 3 && (ks / TB + 1) * TB < nks) fill_batch(ks / TB + 1);
                    RN_SPLITH_MFMA(acc[1][0], sa1, sb0);
                    RN_SPLITH_MFMA(acc[1][1], sa1, sb1);
                } else {
                    const Split8 sa0 = split8(a0), sb0 = split8(b0), sb1 = split8(b1);
                    RN_SPLIT_MFMA(acc[0][0], sa0, sb0);
                    const Split8 sa1 = split8(a1);
                    RN_SPLIT_MFMA(acc[0][1], sa0, sb1);
                    make_offsets();                               // step ks+2 (see the fp32 form below)
                    if ((ks % TB) == 3 && (ks / TB + 1) * TB < nks) fill_batch(ks / TB + 1);
                    RN_SPLIT_MFMA(acc[1][0], sa1, sb0);
                    RN_SPLIT_MFMA(acc[1][1], sa1, sb1);
                }
                rn_wait_dma();
                __syncthreads();
                continue;
            }
            float2 fa[3], fb[3];                              // reads run two k-pairs ahead of the MFMAs that consume them
#pragma unroll
            for (int pre = 0; pre < 2; ++pre) {
                fa[pre] = *reinterpret_cast<const float2 *>(A + 2 * pre * BM);
                fb[pre] = make_float2(B[2 * pre * BN], B[2 * pre * BN + 32]);
            }
#pragma unroll
            for (int kp = 0; kp < WK / 2; ++kp) {
                const int cur = kp % 3, nxt = (kp + 2) % 3;
                if (kp + 2 < WK / 2) {
                    fa[nxt] = *reinterpret_cast<const float2 *>(A + 2 * (kp + 2) * BM);
                    fb[nxt] = make_float2(B[2 * (kp + 2) * BN], B[2 * (kp + 2) * BN + 32]);   // one ds_read2_b32
                }
                if (RELU) { fb[cur].x = fmaxf(fb[cur].x, 0.f); fb[cur].y = fmaxf(fb[cur].y, 0.f); }
                if (CS) { cs.x += fa[cur].x; cs.y += fa[cur].y; }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc[1][1], 0, 0, 0);
                if (kp == WK / 4 - 1) {
                    make_offsets();                           // step ks+2, from the entries read above
                    // next table batch: first read (two steps ahead) at iteration TB*(j+1) - 2; its slot held batch j-1,
                    // last read at iteration TB*j - 3
                    if ((ks % TB) == 3 && (ks / TB + 1) * TB < nks) fill_batch(ks / TB + 1);
                }
            }
            rn_wait_dma();                                    // this wave's loads have landed ...
            __syncthreads();                                  // ... and so have everybody's; buffer `buf` is free again
        }
    };
    if (do_cs) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    if (do_cs) {
        cs.x += __shfl_xor(cs.x, 32);                         // the two k parities of the same channel pair
        cs.y += __shfl_xor(cs.y, 32);
        const int c = m0 + wm * 64 + 2 * (lane & 31);
        if (lane < 32) {
            if (p.cs_slab != nullptr) {                       // one writer per (slice, channel): plain stores
                float *q = p.cs_slab + (int64_t)slice * p.Cout;
                if (c + 0 < p.Cout) q[c + 0] = cs.x;
                if (c + 1 < p.Cout) q[c + 1] = cs.y;
            } else {
                if (c + 0 < p.Cout) atomicAdd(p.colsum + c + 0, cs.x);
                if (c + 1 < p.Cout) atomicAdd(p.colsum + c + 1, cs.y);
            }
        }
    }
    float *out = dw_b;
    if (p.slab != nullptr) out = p.slab + (int64_t)slice * p.slab_stride + (int64_t)batch * p.dw_bstride;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wn * 64 + tn * 32 + (lane & 31);         // contiguous: an atomic instruction covers two full 128-byte row segments
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) + tm;   // tile tm: channels 2*i + tm
                if (row < p.Cout && col < p.Kflat) {
                    const float v = HALF ? acc[tm][tn][e] * us : acc[tm][tn][e];
                    if (p.slab != nullptr) out[(int64_t)row * p.Kpad + col] = v;
                    else RN_WG_ATOMIC(out + (int64_t)row * p.Kpad + col, v);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Split-operand form with the split done ONCE per workgroup (round 3).  In conv_wgrad_kernel<.., SPLIT = true> every wave splits the
// fragments it reads -- both operands are activations -- so each fragment is split by the two waves that share it: 32 values per lane
// and K-step, the largest vector load of any kernel here (knock-out: 23 % of the kernel's time, profiles/r03_big_tile.txt, K8).
// Here a thread loads two float4 of each operand (4 channels of pixels p and p + 8 of the 16-pixel K-step) into registers two steps
// ahead, splits them one step ahead (16 values per lane) and stores the three bf16 terms into plane images [16 pixels][128 columns]
// (conv_wgrad_geom.h: WgradSplitGeom, the bf16 kernel's image); the MFMA phase reads ready operands through gfx950's transposing
// LDS read (ds_read_b64_tr_b16: 8 consecutive pixels of one column per lane, two reads).  128 x 128 tile, 48 KB of LDS, three
// workgroups per CU, same grid / K slices / atomics (or slabs) / column sums as the kernel above.
typedef short s16x4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 wgs_operand(const char *img, unsigned a0, unsigned a1) {
    typedef __attribute__((address_space(3))) s16x4w *lp;
    const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(__attribute__((address_space(3))) char *)(img + a0));
    const s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(__attribute__((address_space(3))) char *)(img + a1));
    typedef short s16x8w __attribute__((ext_vector_type(8)));
    const s16x8w v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// HALF: the fp16 two-term form (RN_FP32_SPLIT3): two planes per operand (a buffer = dY hi, lo, X hi, lo: 16 KB instead of 24), the
// operands scaled by their tensors' powers of two inside the split, three v_mfma_f32_32x32x16_f16 per block, the tile multiplied by the
// two inverse scales before its atomics.
template <bool RELU, bool HALF = false>
__global__ __launch_bounds__(256, 3) void conv_wgrad_once_kernel(const WgradArgs p) {
    using G = WgradSplitGeom;
    constexpr int BM = 128, BN = 128, WK = G::WK, TB = G::TB;
    constexpr int NP = HALF ? 2 : 3;                         // planes per operand
    constexpr int BUFB = 2 * NP * G::IMG;
    static_assert(BUFB <= G::BUF && 8 * BM * 4 <= BUFB, "planes fit; the column-sum reduction fits a buffer");
    __shared__ __attribute__((aligned(16))) char lds[2][BUFB];
    __shared__ int2 pixtab[2][TB * WK];                      // (byte offset of input pixel (ih0, iw0), ih0 | iw0 << 16): 53 KB in all = 3 per CU

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int slice, tile;
    if (p.xcd_map) {
        const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
        slice = (wi / p.tiles) * 8 + xcd;
        tile = wi % p.tiles;
    } else {
        slice = blockIdx.x / p.tiles;
        tile = blockIdx.x % p.tiles;
    }
    if (slice >= p.splits) return;
    const int batch = tile / p.tiles_per_batch;
    tile -= batch * p.tiles_per_batch;
    const float *dy_b = p.dy + (int64_t)batch * p.dy_bstride;
    const float *x_b = p.x + (int64_t)batch * p.x_bstride;
    float *dw_b = p.dw + (int64_t)batch * p.dw_bstride;
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int64_t kbeg = (int64_t)slice * p.per_split;
    const int64_t kend = (kbeg + p.per_split < p.pixels) ? kbeg + p.per_split : p.pixels;
    const int nks = (int)((kend - kbeg + WK - 1) / WK);
    if (nks <= 0) return;
    const int HoWo = p.Ho * p.Wo;
    const int n_first = (int)(kbeg / HoWo);
    const int rel0 = (int)(kbeg - (int64_t)n_first * HoWo);
    const int64_t img = (int64_t)p.Hi * p.Wi * p.Cin;
    int64_t xbytes = ((int64_t)p.N - n_first) * img * 4;
    if (xbytes > 0x7FFFFFFF) xbytes = 0x7FFFFFFF;
    // buffer descriptors: dY = this K slice only (pixels past kend read as zero); X from the first image the slice touches
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dy_b + kbeg * p.ldy), (short)0,
                                                                          (int)(unsigned)((kend - kbeg) * p.ldy * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x_b + (int64_t)n_first * img), (short)0,
                                                                          (int)(unsigned)xbytes, 0x00020000);
    // this thread: 4-channel chunk c of pixels px0 and px0 + 8 of every K-step, for both operands
    const int c = G::chunk(tid), px0 = G::pixel(tid, 0);
    const bool a_col_ok = (m0 + 4 * c) < p.ldy;
    const unsigned a_voff = a_col_ok ? (unsigned)((px0 * p.ldy + m0 + 4 * c) * 4) : 0x80000000u;
    const unsigned a_half = (unsigned)(8 * p.ldy * 4);       // bytes from pixel p to pixel p + 8 (scalar)
    const int jcol = n0 + 4 * c;                             // X: the chunk fixes (tap, first input channel)
    const int tap = jcol / p.Cin;
    const int ci0 = jcol - tap * p.Cin;
    const int fr = tap / p.kw, fs = tap - fr * p.kw;
    const int fr_t = jcol < p.Kflat ? fr : (1 << 24);        // a column past the matrix fails every row test
    const int tap_off = ((fr * p.Wi + fs) * p.Cin + ci0) * 4;
    const unsigned wr0 = (unsigned)G::wr_addr(tid, 0), wr1 = (unsigned)G::wr_addr(tid, 1);

    auto fill_batch = [&](int j) {
        const int rel = rel0 + j * (TB * WK) + tid;
        int2 e = make_int2(0, 0x8000);                       // past the slice: ih0 = -32768 fails the row test (also with fr_t added)
        if (kbeg + (int64_t)j * (TB * WK) + tid < kend) {
            const unsigned n = (unsigned)rel / (unsigned)HoWo;
            const unsigned rem = (unsigned)rel - n * (unsigned)HoWo;
            const unsigned oh = rem / (unsigned)p.Wo, ow = rem - oh * (unsigned)p.Wo;
            const int ih0 = (int)oh * p.stride - p.pad, iw0 = (int)ow * p.stride - p.pad;      // both fit 16 bits (launcher)
            e = make_int2((((int)n * p.Hi + ih0) * p.Wi + iw0) * p.Cin * 4, (ih0 & 0xffff) | (iw0 << 16));
        }
        pixtab[j & 1][tid] = e;
    };
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    struct Regs { f32x4v a[2], b[2]; };
    // loads of K-step ks (all four; offsets of the X half from the pixel table, the dY half from a scalar advance)
    auto load_step = [&](int ks, Regs &r) {
        const int2 *tab = pixtab[(ks / TB) & 1];
        const unsigned so = (unsigned)ks * (unsigned)(WK * 4) * (unsigned)p.ldy;
        r.a[0] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)a_voff, (int)so, 0));
        r.a[1] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)a_voff, (int)(so + a_half), 0));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int2 e = tab[G::tab_index(ks, tid, h)];
            const int ih0 = (int)(short)(e.y & 0xffff), iw0 = e.y >> 16;
            const unsigned ok = (((unsigned)(ih0 + fr_t) < (unsigned)p.Hi) & ((unsigned)(iw0 + fs) < (unsigned)p.Wi)) ? 0xFFFFFFFFu : 0u;
            const unsigned v = ((unsigned)(e.x + tap_off) & ok) | (0x80000000u & ~ok);
            r.b[h] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)v, 0, 0));
        }
    };
    float cs4[4] = {0.f, 0.f, 0.f, 0.f};                     // column sums of this thread's 4 dY channels over its pixels
    const bool do_cs = p.colsum != nullptr && batch == p.colsum_batch && (tile % p.tiles_n) == 0;
    float s_a = 1.f, s_b = 1.f, us = 1.f;                    // HALF: operand scales and the product of their inverses
    if constexpr (HALF) {
        float ua, ub;
        wgrad_scales(p.dy_amax, p.dy_amax_n, s_a, ua);
        wgrad_scales(p.x_amax, p.x_amax_n, s_b, ub);
        us = ua * ub;
    }
    // split the registers of one K-step and store the planes of buffer `buf`
    auto split_store = [&](int buf, const Regs &r) {
        char *B = &lds[buf][0];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned wr = h ? wr1 : wr0;
            float av[4] = {r.a[h][0], r.a[h][1], r.a[h][2], r.a[h][3]};
            float bv[4] = {r.b[h][0], r.b[h][1], r.b[h][2], r.b[h][3]};
            if (RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = fmaxf(bv[j], 0.f);
            }
            if (do_cs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) cs4[j] += av[j];
            }
            if constexpr (HALF) {
                unsigned ah[2], al[2], bh[2], bl[2];
                split_pair_h(av[0], av[1], s_a, ah[0], al[0]);
                split_pair_h(av[2], av[3], s_a, ah[1], al[1]);
                split_pair_h(bv[0], bv[1], s_b, bh[0], bl[0]);
                split_pair_h(bv[2], bv[3], s_b, bh[1], bl[1]);
                *reinterpret_cast<uint2 *>(B + 0 * G::IMG + wr) = make_uint2(ah[0], ah[1]);
                *reinterpret_cast<uint2 *>(B + 1 * G::IMG + wr) = make_uint2(al[0], al[1]);
                *reinterpret_cast<uint2 *>(B + 2 * G::IMG + wr) = make_uint2(bh[0], bh[1]);
                *reinterpret_cast<uint2 *>(B + 3 * G::IMG + wr) = make_uint2(bl[0], bl[1]);
                continue;
            }
            unsigned ah[2], am[2], al[2], bh[2], bm[2], bl[2];
            split_pair(av[0], av[1], ah[0], am[0], al[0]);
            split_pair(av[2], av[3], ah[1], am[1], al[1]);
            split_pair(bv[0], bv[1], bh[0], bm[0], bl[0]);
            split_pair(bv[2], bv[3], bh[1], bm[1], bl[1]);
            *reinterpret_cast<uint2 *>(B + 0 * G::IMG + wr) = make_uint2(ah[0], ah[1]);
            *reinterpret_cast<uint2 *>(B + 1 * G::IMG + wr) = make_uint2(am[0], am[1]);
            *reinterpret_cast<uint2 *>(B + 2 * G::IMG + wr) = make_uint2(al[0], al[1]);
            *reinterpret_cast<uint2 *>(B + 3 * G::IMG + wr) = make_uint2(bh[0], bh[1]);
            *reinterpret_cast<uint2 *>(B + 4 * G::IMG + wr) = make_uint2(bm[0], bm[1]);
            *reinterpret_cast<uint2 *>(B + 5 * G::IMG + wr) = make_uint2(bl[0], bl[1]);
        }
    };
    unsigned fa[2][2], fb[2][2];                             // transposing-read addresses: [32-column sub-tile][read 0 / 1]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            fa[t][rd] = (unsigned)G::tr_addr(wm, t, rd, lane);
            fb[t][rd] = (unsigned)(NP * G::IMG + G::tr_addr(wn, t, rd, lane));
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e_ = 0; e_ < 16; ++e_) acc[i][j][e_] = 0.f;

    fill_batch(0);
    __syncthreads();
    Regs r0, r1;
    load_step(0, r0);
    split_store(0, r0);
    load_step(1, r0);                                        // step 1 waits in registers for iteration 0
    __syncthreads();
    // One K-step: `cur` holds step ks + 1 (loaded an iteration ago), `nxt` receives step ks + 2 (steps past the slice load zeros:
    // the dY descriptor ends at kend and the table marks those pixels invalid).
    auto k_step = [&](int ks, int buf, Regs &cur, Regs &nxt) {
        const char *S = &lds[buf][0];
        if constexpr (HALF) {
            SplitH8 sa[2], sb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                sa[t].h = __builtin_bit_cast(f16x8, wgs_operand(S, fa[t][0], fa[t][1]));
                sa[t].l = __builtin_bit_cast(f16x8, wgs_operand(S + G::IMG, fa[t][0], fa[t][1]));
                sb[t].h = __builtin_bit_cast(f16x8, wgs_operand(S, fb[t][0], fb[t][1]));
                sb[t].l = __builtin_bit_cast(f16x8, wgs_operand(S + G::IMG, fb[t][0], fb[t][1]));
            }
            split_store(buf ^ 1, cur);
            load_step(ks + 2, nxt);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) RN_SPLITH_MFMA(acc[tm][tn], sa[tm], sb[tn]);
        } else {
        Split8 sa[2], sb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sa[t].h = wgs_operand(S, fa[t][0], fa[t][1]);
            sa[t].m = wgs_operand(S + G::IMG, fa[t][0], fa[t][1]);
            sa[t].l = wgs_operand(S + 2 * G::IMG, fa[t][0], fa[t][1]);
            sb[t].h = wgs_operand(S, fb[t][0], fb[t][1]);
            sb[t].m = wgs_operand(S + G::IMG, fb[t][0], fb[t][1]);
            sb[t].l = wgs_operand(S + 2 * G::IMG, fb[t][0], fb[t][1]);
        }
        split_store(buf ^ 1, cur);
        load_step(ks + 2, nxt);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) RN_SPLIT_MFMA(acc[tm][tn], sa[tm], sb[tn]);
        }
        // table batch b is first read for step b * TB (loaded in iteration b * TB - 2); its slot held batch b - 2, last read in
        // iteration (b - 1) * TB - 3: fill it in iteration (b - 1) * TB + 4
        if ((ks % TB) == 4 && (ks / TB + 1) * TB < nks + 2) fill_batch(ks / TB + 1);
        __syncthreads();
    };
    for (int ks = 0; ks < nks; ks += 2) {
        k_step(ks, 0, r0, r1);
        if (ks + 1 < nks) k_step(ks + 1, 1, r1, r0);
    }
    if (do_cs) {                                             // 8 threads hold partial sums of the same 4 channels: through LDS
        float (*csred)[BM] = reinterpret_cast<float (*)[BM]>(&lds[0][0]);     // (the planes are done with: the loop ended in a barrier)
#pragma unroll
        for (int j = 0; j < 4; ++j) csred[tid >> 5][4 * c + j] = cs4[j];
        __syncthreads();
        if (tid < BM) {
            float s_ = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s_ += csred[k][tid];
            const int ch = m0 + tid;
            if (ch < p.Cout) {
                if (p.cs_slab != nullptr) p.cs_slab[(int64_t)slice * p.Cout + ch] = s_;
                else atomicAdd(p.colsum + ch, s_);
            }
        }
    }
    float *out = dw_b;
    if (p.slab != nullptr) out = p.slab + (int64_t)slice * p.slab_stride + (int64_t)batch * p.dw_bstride;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wn * 64 + tn * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < p.Cout && col < p.Kflat) {
                    const float v = HALF ? acc[tm][tn][e] * us : acc[tm][tn][e];
                    if (p.slab != nullptr) out[(int64_t)row * p.Kpad + col] = v;
                    else RN_WG_ATOMIC(out + (int64_t)row * p.Kpad + col, v);
                }
            }
        }
}

// Deterministic form, second pass: dw[b][row][col] += slab[0] + slab[1] + ... in slice order (one thread per element, col
// fastest: coalesced over every slab), colsum[c] likewise from the column-sum slabs.
__global__ __launch_bounds__(256) void wgrad_combine_kernel(float *__restrict__ dw, const float *__restrict__ slab, int64_t slab_stride,
                                                            int splits, int nbatch, int64_t dw_bstride, int Cout, int Kflat, int Kpad,
                                                            float *__restrict__ colsum, const float *__restrict__ cs_slab) {
    const int64_t per = (int64_t)Cout * Kflat, total = per * nbatch;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) {
        const int64_t b = i / per, r = i - b * per;
        const int row = (int)(r / Kflat), col = (int)(r - (int64_t)row * Kflat);
        const int64_t e = b * dw_bstride + (int64_t)row * Kpad + col;
        float v = dw[e];
        for (int s = 0; s < splits; ++s) v += slab[(int64_t)s * slab_stride + e];
        dw[e] = v;
    }
    if (colsum != nullptr && i < Cout) {
        float v = colsum[i];
        for (int s = 0; s < splits; ++s) v += cs_slab[(int64_t)s * Cout + i];
        colsum[i] = v;
    }
}

static int wgrad_launch(const float *dy, int ldy, const float *x, float *dw, float *colsum, int nbatch,
                        int64_t dy_bstride, int64_t x_bstride, int64_t dw_bstride, int colsum_batch, int N, int Hi,
                        int Wi, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                        int in_relu, void *stream, bool det, void *workspace, int64_t workspace_bytes, int64_t *need_bytes,
                        const void *dy_amax = nullptr, int dy_amax_n = 0, const void *x_amax = nullptr, int x_amax_n = 0);

extern "C" int rn_conv_wgrad_batched(const float *dy, int ldy, const float *x, float *dw, float *colsum, int nbatch,
                                     int64_t dy_bstride, int64_t x_bstride, int64_t dw_bstride, int colsum_batch, int N, int Hi,
                                     int Wi, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                     int in_relu, const void *dy_amax, int dy_amax_n, const void *x_amax, int x_amax_n, void *stream) {
    return wgrad_launch(dy, ldy, x, dw, colsum, nbatch, dy_bstride, x_bstride, dw_bstride, colsum_batch, N, Hi, Wi, Cin, Ho, Wo, Cout,
                        kh, kw, stride, pad, in_relu, stream, false, nullptr, 0, nullptr, dy_amax, dy_amax_n, x_amax, x_amax_n);
}

// Fixed-order reduction (RN_OPT_DETERMINISTIC): same kernel, K slices store into slabs of `workspace`, one ordered combine.
extern "C" int rn_conv_wgrad_batched_det(const float *dy, int ldy, const float *x, float *dw, float *colsum, int nbatch,
                                         int64_t dy_bstride, int64_t x_bstride, int64_t dw_bstride, int colsum_batch, int N, int Hi,
                                         int Wi, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                         int in_relu, const void *dy_amax, int dy_amax_n, const void *x_amax, int x_amax_n,
                                         void *workspace, int64_t workspace_bytes, void *stream) {
    if (workspace == nullptr) return RN_EINVAL;
    return wgrad_launch(dy, ldy, x, dw, colsum, nbatch, dy_bstride, x_bstride, dw_bstride, colsum_batch, N, Hi, Wi, Cin, Ho, Wo, Cout,
                        kh, kw, stride, pad, in_relu, stream, true, workspace, workspace_bytes, nullptr, dy_amax, dy_amax_n, x_amax, x_amax_n);
}

// Bytes of workspace rn_conv_wgrad_batched_det needs for this problem (slices x (result image + Cout column sums)).
extern "C" int64_t rn_conv_wgrad_det_workspace_bytes(int ldy, int nbatch, int64_t dw_bstride, int N, int Hi, int Wi, int Cin, int Ho,
                                                     int Wo, int Cout, int kh, int kw) {
    int64_t need = 0;
    const int rc = wgrad_launch(nullptr, ldy, nullptr, nullptr, nullptr, nbatch, 0, 0, dw_bstride, 0, N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw,
                                1, 0, 0, nullptr, true, nullptr, 0, &need);
    return rc == RN_OK ? need : -1;
}

static int wgrad_launch(const float *dy, int ldy, const float *x, float *dw, float *colsum, int nbatch,
                        int64_t dy_bstride, int64_t x_bstride, int64_t dw_bstride, int colsum_batch, int N, int Hi,
                        int Wi, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                        int in_relu, void *stream, bool det, void *workspace, int64_t workspace_bytes, int64_t *need_bytes,
                        const void *dy_amax, int dy_amax_n, const void *x_amax, int x_amax_n) {
    if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin < 4 || (Cin & 3) || (ldy & 3) || ldy < Cout)
        return RN_EINVAL;
    if (nbatch < 1 || nbatch > 4096 || dy_bstride < 0 || x_bstride < 0 || dw_bstride < 0 || colsum_batch < 0 || colsum_batch >= nbatch)
        return RN_EINVAL;
    WgradArgs a;
    a.dy_bstride = dy_bstride; a.x_bstride = x_bstride; a.dw_bstride = dw_bstride; a.colsum_batch = colsum_batch;
    a.dy = dy; a.x = x; a.dw = dw; a.colsum = colsum; a.ldy = ldy;
    a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad; a.in_relu = in_relu;
    a.Kflat = kh * kw * Cin;
    a.Kpad = (a.Kflat + 31) / 32 * 32;
    a.pixels = (int64_t)N * Ho * Wo;
    // tile shape: 64 x 256 for few output channels, 256 x 64 for a short packed row (1x1 from 64 channels), else 128 x 128
    const int shape = Cout <= 64 ? 0 : (a.Kflat <= 64 ? 1 : 2);
    const int BM = shape == 0 ? 64 : (shape == 1 ? 256 : 128), BN = shape == 0 ? 256 : (shape == 1 ? 64 : 128);
    const int tiles_m = (Cout + BM - 1) / BM;
    a.tiles_n = (a.Kflat + BN - 1) / BN;
    a.tiles_per_batch = tiles_m * a.tiles_n;
    const int tiles = a.tiles_per_batch * nbatch;
    // enough K slices for ~2048 workgroups -- two rounds of what the chip holds (4 per CU at 40 KB of LDS); measured per
    // training step: 512 -> 66.0 ms, 1024 -> 57.3, 1536 -> 55.2, 2048 -> 54.2, 4096 -> 54.9, 8192 -> 55.1 -- each at
    // least 512 pixels long
    static const int wgs_native = getenv("RN_WGRAD_WGS") ? atoi(getenv("RN_WGRAD_WGS")) : 2048;
    static const int wgs_split = getenv("RN_WGRAD_SPLIT_WGS") ? atoi(getenv("RN_WGRAD_SPLIT_WGS")) : 1536;   // split kernels: three per CU -> two rounds of 768 (measured 768 / 1024 / 1536 / 2048 / 3072: 21.8 / 21.5 / 20.6 / 21.0 / 21.5 ms per step)
    const int wg_target = rn_get_fp32_mfma() != RN_FP32_NATIVE ? wgs_split : wgs_native;
    int64_t splits = (wg_target + tiles - 1) / tiles;
    const int64_t max_splits = (a.pixels + 16 * WK_MAX - 1) / (16 * WK_MAX);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    a.xcd_map = tiles >= 16 && splits > 8 && splits + 7 <= max_splits;
    if (a.xcd_map) splits = (splits + 7) / 8 * 8;            // whole slices per XCD: equal shares for the 8
    // Buffer-load addressing (see the kernel): the dY bytes of one K slice and the span of input images a slice can
    // touch must each stay below 2 GiB; more slices make both smaller.
    const int64_t HoWo = (int64_t)Ho * Wo, img_bytes = (int64_t)Hi * Wi * Cin * 4;
    for (;;) {
        a.per_split = ((a.pixels + splits - 1) / splits + WK_MAX - 1) / WK_MAX * WK_MAX;
        const int64_t span_imgs = (a.per_split + HoWo - 2) / HoWo + 1;
        if ((a.per_split + WK_MAX) * ldy * 4 <= 0x7FFFFFFF && span_imgs * img_bytes <= 0x7FFFFFFF) break;
        if (a.per_split <= WK_MAX || splits >= 65535) return RN_EINVAL;   // a single image of > 2 GiB
        splits = splits * 2 > 65535 ? 65535 : splits * 2;
        a.xcd_map = 0;
    }
    splits = (a.pixels + a.per_split - 1) / a.per_split;
    a.tiles = tiles;
    a.splits = (int)splits;
    a.slab = a.cs_slab = nullptr;
    a.slab_stride = 0;
    if (det) {
        a.slab_stride = nbatch > 1 ? (int64_t)nbatch * dw_bstride : (int64_t)Cout * a.Kpad;
        const int64_t need = splits * (a.slab_stride + Cout) * (int64_t)sizeof(float);
        if (need_bytes != nullptr) { *need_bytes = need; return RN_OK; }
        if (workspace_bytes < need) return RN_EINVAL;
        a.slab = reinterpret_cast<float *>(workspace);
        a.cs_slab = a.slab + splits * a.slab_stride;
    }
    const dim3 grid((unsigned)(tiles * (a.xcd_map ? (splits + 7) / 8 * 8 : splits)));
    // 16-pixel K-steps: 40-48 KiB of LDS, four / three workgroups per CU (measured: +2..4 % over 32-pixel steps at two per CU
    // on the mid-size layers, equal on the largest; the 64 x 256 tile does not fit twice at 32).
    const bool split = rn_get_fp32_mfma() != RN_FP32_NATIVE;
    // RN_FP32_SPLIT3 with both amax words: the fp16 two-term kernels; without them (a caller of the plain entry point) the three-term ones
    const bool half = rn_get_fp32_mfma() == RN_FP32_SPLIT3 && dy_amax != nullptr && x_amax != nullptr && dy_amax_n != 0 && x_amax_n != 0;
    a.dy_amax = dy_amax;
    a.x_amax = x_amax;
    a.dy_amax_n = dy_amax_n;
    a.x_amax_n = x_amax_n;
#define RN_WGRAD_LAUNCH(WM_, WN_)                                                                                        \
    do {                                                                                                                 \
        if (half) {                                                                                                      \
            if (in_relu) hipLaunchKernelGGL((conv_wgrad_kernel<WM_, WN_, true, 16, 3, true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);  \
            else hipLaunchKernelGGL((conv_wgrad_kernel<WM_, WN_, false, 16, 3, true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);         \
        } else if (split) {                                                                                              \
            if (in_relu) hipLaunchKernelGGL((conv_wgrad_kernel<WM_, WN_, true, 16, 3, true>), grid, dim3(256), 0, (hipStream_t)stream, a);  \
            else hipLaunchKernelGGL((conv_wgrad_kernel<WM_, WN_, false, 16, 3, true>), grid, dim3(256), 0, (hipStream_t)stream, a);         \
        } else if (in_relu) hipLaunchKernelGGL((conv_wgrad_kernel<WM_, WN_, true, 16, 3>), grid, dim3(256), 0, (hipStream_t)stream, a);  \
        else hipLaunchKernelGGL((conv_wgrad_kernel<WM_, WN_, false, 16, 3>), grid, dim3(256), 0, (hipStream_t)stream, a);         \
    } while (0)
    // RN_OPT_WGRAD_ONCE = 0: the per-wave split for the 128 x 128 tile too (A/B)
    const bool once = split && shape == 2 && rn_get_option(RN_OPT_WGRAD_ONCE) != 0 && Hi + pad < 32000 && Wi + pad < 32000;
    if (once && half) {
        if (in_relu) hipLaunchKernelGGL((conv_wgrad_once_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv_wgrad_once_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    } else if (once) {
        if (in_relu) hipLaunchKernelGGL((conv_wgrad_once_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv_wgrad_once_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    } else if (shape == 0) RN_WGRAD_LAUNCH(1, 4);
    else if (shape == 1) RN_WGRAD_LAUNCH(4, 1);
    else RN_WGRAD_LAUNCH(2, 2);
#undef RN_WGRAD_LAUNCH
    RN_LAUNCH_CHECK();
    if (det) {
        const int64_t total = (int64_t)nbatch * Cout * a.Kflat;
        hipLaunchKernelGGL(wgrad_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dw,
                           (const float *)a.slab, a.slab_stride, a.splits, nbatch, dw_bstride, Cout, a.Kflat, a.Kpad, colsum,
                           (const float *)a.cs_slab);
        RN_LAUNCH_CHECK();
    }
    return RN_OK;
}

extern "C" int rn_conv_wgrad(const float *dy, int ldy, const float *x, float *dw, float *colsum, int N, int Hi, int Wi,
                             int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int in_relu,
                             void *stream) {
    return rn_conv_wgrad_batched(dy, ldy, x, dw, colsum, 1, 0, 0, 0, 0, N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad,
                                 in_relu, nullptr, 0, nullptr, 0, stream);
}
