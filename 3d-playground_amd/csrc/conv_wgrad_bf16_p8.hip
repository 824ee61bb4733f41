// bf16 weight gradient, eight-wave 256 x 256 tile over 64-pixel K-tiles with a phased K loop (round 4) -- conv_bf16_p8.hip's structure
// for dW[co][(tap, ci)] += sum over pixels of dY[pixel][co] * X[pixel + tap][ci] (the gradient of every stride-1 same-size convolution
// of D/model.py:59-205 whose channel counts are multiples of 256: the head towers, the pyramid's 3x3 layers, layer3 / layer4's conv2 and
// the wide 1x1 layers).  conv_bf16.hip's weight gradient (128 x 128 tile, a barrier per 32 pixels = 8 MFMAs per wave) moves twice the
// bytes per MAC through the direct-to-LDS path and sits at 600-810 TFLOP/s on those layers (profiles/r04_bf16_step_by_shape.txt).
//
//   * tile = 256 output channels x 256 columns of ONE filter tap (Cin % 256 == 0); a workgroup reduces one slice of the pixels and adds
//     its fp32 tile with atomics, as conv_bf16.hip's kernel does (the slices of a tile are spread over the grid: one round of the CUs);
//   * operands are staged as they lie, [64 pixels][128 columns] images of 256-byte rows (two per operand and K-tile), 16-byte chunks
//     XOR-permuted by WgradBf16Geom::fx -- the image conv_bf16.hip's kernel reads with ds_read_b64_tr_b16 (conv_wgrad_geom.h; 64 rows
//     instead of 32: fx has period 16);
//   * 2 x 4 waves, each 128 channels x 64 columns on v_mfma_f32_32x32x16_bf16 (4 x 2 accumulators); a K-tile is four phases of 8 MFMAs
//     (16 pixels each); a phase's operands are read during the phase before, the staging of the K-tile after next is issued between the
//     MFMAs of the last phase, ONE s_waitcnt vmcnt(0) + s_barrier per K-tile;
//   * the padding: a stride-1 same-size layer's pixels are one flat sequence, so a tap is a uniform shift of the lane's byte offset and a
//     lane whose shifted pixel leaves the image sends the out-of-range offset (zero-fill); the lane keeps (row, column) of its two
//     pixels and advances them by 64 pixels per K-tile without divisions.
// Column sums of dY (the bias gradient) come from the tap-0 workgroups' operands, as in conv_bf16.hip.
//
// RESULT (profiles/r04_wgrad_p8_knockouts.txt): correct (tests/test_gpu_conv_bf16_p8.py) and slower than the kernel it was meant to
// replace -- head tower 3x3 256 -> 256 at 8 x 135 x 240: 0.51 ms against 0.40.  Knocked out one by one: MFMAs alone 0.31 ms (one workgroup
// of eight waves per CU on v_mfma_f32_32x32x16_bf16 reaches ~1 000 TFLOP/s on random operands), + the transposing operand reads 0.35,
// + the staging 0.46, + 64 MB of tile atomics that nothing overlaps (one round of workgroups ends together) 0.51: the parts add up
// instead of hiding each other, and the 128 x 128 kernel's three independent workgroups per CU hide them.  Not selected by default
// (conv_bf16.hip: rn_conv_wgrad_bf16); kept with its tests as the measured reference for the next attempt (16x16x32 MFMAs, which
// ran 1.4x faster in conv_bf16_p8.hip's loop, need one v_xor per transposing read on this image).
#include <stdlib.h>

#include "common.h"
#include "conv_wgrad_geom.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradP8Args {
    const __bf16 *dy, *x;
    float *dw, *colsum;
    int ldy, H, W, Cin, Cout, kw, pad;
    int Kpad, tiles_n, tiles, splits;
    int64_t pixels, per_split;                   // per_split: a multiple of 64
};

#ifndef WP_ABL
#define WP_ABL 0                                 // knock-outs for profiles/ (bits): 1 no atomics, 4 no staging after the prologue
#endif
constexpr int WP_IMG = 64 * 256;                 // bytes of one [64 pixels][128 columns] image
constexpr int WP_BUF = 4 * WP_IMG;               // a K-tile: dY columns 0-127, 128-255, X columns 0-127, 128-255
constexpr int WP_LDS = 2 * WP_BUF;               // 128 KB (dynamic)

// One 32x32x16 operand: lane l <- 8 consecutive pixels 16 kh + 8 (l >> 5) + 0..7 of column (l & 31) of a 32-column sub-tile, by two
// ds_read_b64_tr_b16 (conv_bf16.hip: wg_operand; the addresses are WgradBf16Geom::tr_addr(.., kh = 0, ..) + kh * 16 rows).
__device__ __forceinline__ bf16x8 wp_operand(const char *img, unsigned rd0, unsigned rd1, int kh) {
    typedef __attribute__((address_space(3))) s16x4 *lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(__attribute__((address_space(3))) char *)(img + rd0 + kh * 16 * 256));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(__attribute__((address_space(3))) char *)(img + rd1 + kh * 16 * 256));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512, 2) void conv_wgrad_bf16_p8_kernel(const WgradP8Args p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave >> 2, wc = wave & 3;     // channels wr * 128, columns wc * 64 of the tile
    // workgroup id -> (tile, pixel slice): the tiles of a slice (its taps: the same dY, overlapping X) on ONE XCD (conv_bf16.hip)
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
    const int slice = (wi / p.tiles) * 8 + xcd, tile = wi % p.tiles;
    if (slice >= p.splits) return;               // padding of the last group of 8 slices
    const int m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * 256;
    const int tap = n0 / p.Cin, ci0 = n0 - tap * p.Cin;
    const int fr = tap / p.kw, dh = fr - p.pad, dw_ = tap - fr * p.kw - p.pad;
    const int64_t kbeg = (int64_t)slice * p.per_split;
    const int64_t kend = kbeg + p.per_split < p.pixels ? kbeg + p.per_split : p.pixels;
    const int npix = (int)(kend - kbeg);
    if (npix <= 0) return;
    const int nkt = (npix + 63) / 64;
    const int Cin = p.Cin, W = p.W, H = p.H;

    // descriptors: dY = this slice's pixels only (past kend: out of range by themselves); X from `halo` pixels in front of the slice
    const int halo = p.pad * W + p.pad;
    const int64_t base_row = kbeg > halo ? kbeg - halo : 0;
    const int64_t xb = (p.pixels - base_row) * Cin * 2;
    const v4i32 rs_a = make_rsrc(p.dy + kbeg * p.ldy, (unsigned)((int64_t)npix * p.ldy * 2));
    const v4i32 rs_b = make_rsrc(p.x + base_row * Cin, (unsigned)(xb > 0x7FFFFFFF ? 0x7FFFFFFF : xb));
    const unsigned lds0 = lds_addr(lds);

    // ---- staging.  Instruction i of this wave fills rows 32 (i & 1) + 4 wave + (lane >> 4) of image i >> 1 (0, 1: dY, 2, 3: X); the
    // lane fills chunk position lane & 15 and fetches logical chunk (lane & 15) ^ fx(row) -- fx does not change with i.
    const int row0 = 4 * wave + (lane >> 4);
    const int chunk = (lane & 15) ^ WgradBf16Geom::fx(row0);
    const unsigned voff_a = (unsigned)((row0 * p.ldy + m0 + 8 * chunk) * 2);
    const unsigned voff_b = (unsigned)((((int)(kbeg - base_row) + row0) * Cin + ci0 + 8 * chunk) * 2);
    // the lane's two pixels (rows row0 and row0 + 32 of the K-tile being staged): row / column in their image, advanced by 64 per K-tile
    int oh[2], ow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned rem = (unsigned)(kbeg + row0 + 32 * j) % (unsigned)(H * W);      // pixels < 2^31 (launcher)
        oh[j] = (int)(rem / (unsigned)W);
        ow[j] = (int)(rem - (unsigned)oh[j] * (unsigned)W);
    }
    const int q64 = 64 / W, r64 = 64 - q64 * W;  // q64 + 1 < H (launcher)
    auto uni = [](const v4i32 r) {
        v4i32 o;
        o.x = __builtin_amdgcn_readfirstlane(r.x); o.y = __builtin_amdgcn_readfirstlane(r.y);
        o.z = __builtin_amdgcn_readfirstlane(r.z); o.w = __builtin_amdgcn_readfirstlane(r.w);
        return o;
    };
    int kt_dma = 0;                              // the K-tile the next dma() calls stage (uniform)
    auto dma = [&](const int i, const int buf) {
        const unsigned dst = lds0 + (unsigned)(buf * WP_BUF + i * 8192 + wave_u * 1024);
        const int j = i & 1, half = (i >> 1) & 1;
        if (i < 4) {
            dma16(uni(rs_a), dst, voff_a, (unsigned)__builtin_amdgcn_readfirstlane(((kt_dma * 64 + 32 * j) * p.ldy + 128 * half) * 2));
        } else {
            const int sh = __builtin_amdgcn_readfirstlane(((dh * W + dw_ + kt_dma * 64 + 32 * j) * Cin + 128 * half) * 2);
            const bool ok = (kt_dma * 64 + 32 * j + row0 < npix) & ((unsigned)(oh[j] + dh) < (unsigned)H) & ((unsigned)(ow[j] + dw_) < (unsigned)W);
            dma16(uni(rs_b), dst, ok ? voff_b + (unsigned)sh : 0x80000000u, 0u);
        }
    };
    auto next_ktile = [&]() {
        ++kt_dma;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ow[j] += r64; oh[j] += q64;
            if (ow[j] >= W) { ow[j] -= W; ++oh[j]; }
            if (oh[j] >= H) oh[j] -= H;
        }
    };

    // ---- operand read addresses (bytes within an image; kh adds 16 rows): dY sub-tiles s = 0..3 (32 channels each) of image wr,
    // X sub-tiles t = 0, 1 of the wave's 64 columns in image 2 + (wc >> 1)
    unsigned fa[4][2], fb[2][2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[s][rd] = (unsigned)(wr * WP_IMG + WgradBf16Geom::tr_addr(s >> 1, s & 1, rd, 0, lane));
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[t][rd] = (unsigned)((2 + (wc >> 1)) * WP_IMG + WgradBf16Geom::tr_addr(wc & 1, t, rd, 0, lane));
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][t][e] = 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_cs = p.colsum != nullptr && (tile % p.tiles_n) == 0 && wc == 0;

    bf16x8 A0[4], B0[2], A1[4], B1[2];           // the operands of even / odd phases
    auto read_ops = [&](bf16x8 (&A)[4], bf16x8 (&B)[2], const char *S, const int kh) {
        if ((WP_ABL & 8) && S != lds + 1) return;             // knock-out: no operand reads in the loop (the prologue's stay)
#pragma unroll
        for (int t = 0; t < 2; ++t) B[t] = wp_operand(S, fb[t][0], fb[t][1], kh);
#pragma unroll
        for (int s = 0; s < 4; ++s) A[s] = wp_operand(S, fa[s][0], fa[s][1], kh);
    };
    auto col_sums = [&](const bf16x8 (&A)[4]) {
        if (do_cs) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[s] += (float)A[s][e];
        }
    };
    auto mma = [&](const bf16x8 (&A)[4], const bf16x8 (&B)[2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s], B[t], acc[s][t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: K-tiles 0 and 1 (all 8 instructions each); wait for tile 0, read its first operands
#pragma unroll
    for (int i = 0; i < 8; ++i) dma(i, 0);
    next_ktile();
    if (nkt > 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(i, 1);
        next_ktile();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    read_ops(A0, B0, (WP_ABL & 8) ? lds + 1 : lds, 0);
    if (WP_ABL & 8) { read_ops(A1, B1, lds + 1, 1); }

    // ---- main loop: K-tile t in buffer t & 1, phases kh = 0..3 on operand sets 0, 1, 0, 1.  Before the barrier every read of the
    // buffer has RETURNED (the reads of phase 3 are issued in phase 2 and not consumed before it: the explicit lgkmcnt) and K-tile
    // t + 1 has landed; after it the buffer is restaged for K-tile t + 2 between the MFMAs of phase 3.
    for (int t = 0; t < nkt; ++t) {
        const char *S = lds + (t & 1) * WP_BUF;
        const char *Sn = lds + ((t + 1) & 1) * WP_BUF;
        read_ops(A1, B1, S, 1);
        col_sums(A0);
        mma(A0, B0);
        read_ops(A0, B0, S, 2);
        col_sums(A1);
        mma(A1, B1);
        read_ops(A1, B1, S, 3);
        col_sums(A0);
        mma(A0, B0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (t + 1 < nkt) read_ops(A0, B0, Sn, 0);
        col_sums(A1);
        const bool stage = t + 2 < nkt && !(WP_ABL & 4);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                acc[s][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[s], B1[tt], acc[s][tt], 0, 0, 0);
                if (stage) dma(2 * s + tt, t & 1);
            }
        __builtin_amdgcn_s_setprio(0);
        if (stage) next_ktile();
    }

    if (WP_ABL & 1) {                                          // keep the accumulators alive, store next to nothing
        float tsum = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) tsum += acc[s][t][0] + acc[s][t][15];
        if (tsum == 1.2345e-30f) p.dw[0] = tsum;
        return;
    }
    // ---- epilogue: column sums, then the fp32 tile by atomics (acc[s][t][e]: channel 32 s + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column lane & 31)
    if (do_cs) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            cs[s] += __shfl_xor(cs[s], 32);                   // the two 8-pixel groups of the same channel
            if (lane < 32) atomicAdd(p.colsum + m0 + wr * 128 + s * 32 + lane, cs[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = n0 + wc * 64 + t * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * 128 + s * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                atomicAdd(p.dw + (int64_t)row * p.Kpad + col, acc[s][t][e]);
            }
        }
}

// What the kernel computes: stride 1, output plane = input plane (pad = (k - 1) / 2, square filter), both channel counts multiples of
// 256, at least two image rows per 64 pixels' advance (the lane's row / column update assumes at most one wrap of each).
bool rn_wgrad_bf16_p8_legal(int ldy, int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad) {
    if (stride != 1 || Hi != Ho || Wi != Wo || kh != kw || 2 * pad != kh - 1) return false;
    if ((Cin & 255) || (Cout & 255) || ldy < Cout || (ldy & 7)) return false;
    if (64 / Wi + 2 > Hi) return false;
    const int64_t pixels = (int64_t)N * Hi * Wi;
    if (pixels + 128 > 0x7fffffffLL || (int64_t)kh * kw * Cin > 0x7fffffLL) return false;
    return true;
}

int rn_wgrad_bf16_p8_launch(const void *dy, int ldy, const void *x, float *dw, float *colsum, int N, int H, int W, int Cin, int Cout,
                            int k, int pad, hipStream_t stream) {
    static const hipError_t attr = hipFuncSetAttribute((const void *)conv_wgrad_bf16_p8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS);
    if (attr != hipSuccess) return (int)attr;
    static const int n_cu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    WgradP8Args a;
    a.dy = reinterpret_cast<const __bf16 *>(dy); a.x = reinterpret_cast<const __bf16 *>(x); a.dw = dw; a.colsum = colsum;
    a.ldy = ldy; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.kw = k; a.pad = pad;
    const int Kflat = k * k * Cin;
    a.Kpad = (Kflat + 31) / 32 * 32;
    a.tiles_n = Kflat / 256;
    a.tiles = (Cout / 256) * a.tiles_n;
    a.pixels = (int64_t)N * H * W;
    // one resident round: a workgroup per CU (RN_WGRAD_P8_WGS overrides the target); a slice is at least 4 K-tiles
    static const int target_env = [] { const char *e = getenv("RN_WGRAD_P8_WGS"); return e ? atoi(e) : 0; }();
    const int target = target_env > 0 ? target_env : n_cu;
    int64_t splits = target / a.tiles;
    const int64_t max_splits = (a.pixels + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    for (;;) {
        a.per_split = ((a.pixels + splits - 1) / splits + 63) / 64 * 64;
        if ((a.per_split + 64) * ldy * 2 <= 0x7FFFFFFF && (a.per_split + 2 * ((int64_t)pad * W + pad) + 128) * Cin * 2 <= 0x7FFFFFFF) break;
        splits *= 2;
        if (splits > 65535) return RN_EINVAL;
    }
    splits = (a.pixels + a.per_split - 1) / a.per_split;
    a.splits = (int)splits;
    const int64_t grid = (int64_t)a.tiles * ((splits + 7) / 8 * 8);
    hipLaunchKernelGGL(conv_wgrad_bf16_p8_kernel, dim3((unsigned)grid), dim3(512), WP_LDS, stream, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
