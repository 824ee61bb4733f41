// Weight gradient of the convolutions on the fp32 matrix cores.
//
// The reference obtains dW from torch autograd (cuDNN wgrad kernels) for every nn.Conv2d of D/model.py; here it
// is one GEMM per layer:  dW[co][r][s][ci] = sum over output pixels of dY[pixel][co] * X[pixel shifted by the
// tap][ci], i.e.  M = Cout,  N = kh*kw*Cin (the packed-weight row, so a tile may span several taps of a
// narrow layer),  K = N*Ho*Wo pixels.
//
// Both operands are pixel-major in memory (NHWC: one pixel = one contiguous channel vector) and are staged into LDS
// exactly as they lie -- [32 pixels][channels], plain 16-byte copies.  The MFMA wants one operand value per lane,
// A[row = lane&31][k = lane>>5]: a lane reads two adjacent channels of pixel k with one ds_read_b64 and feeds them to
// two different 32-row tiles (tile tm holds channels 2*i + tm), so no transposition is ever done; the permutation is
// undone by the epilogue's index arithmetic.  32 ds_read_b64 against 64 MFMAs (64 cycles each) per wave per K-step,
// every fragment address an immediate.
//
// K is split over gridDim.y; partial tiles are added with fp32 atomics (two 128-byte row segments per wave
// instruction, the full-rate shape) into a buffer the caller zeroes -- the five pyramid levels of a shared head
// accumulate into the same buffer.  All tiles of one K-slice run on the same XCD so they stream the same dY /
// X slabs through one L2.
//
// The column sums of dY (bias / batch-norm beta gradients, and the mean term of the gamma gradient) ride along:
// the workgroups of N-tile 0 add up the dY chunks they stage anyway and finish with one atomic per channel.
//
// Roofline: MFMA (fp32 157.3 TF).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define WK 32                    // pixels per K-step
#define TB 8                     // K-steps per pixel-table batch (TB * WK = 256 = one entry per thread)
typedef int v4i32 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const float *dy, *x;
    float *dw;
    float *colsum;               // [Cout] accumulated sum over pixels of dy, or NULL
    int ldy, N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad, in_relu;
    int Kflat, Kpad;             // kh*kw*Cin and its round-up to 32
    int tiles_n;                 // number of N tiles
    int tiles, splits;           // tiles_m * tiles_n, K slices
    int xcd_map;                 // 1: whole K slices per XCD (see the kernel)
    int64_t pixels, per_split;   // K extent and K per grid.y slice (multiple of WK)
};

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int CA = BM / 4, CB = BN / 4;                  // 4-channel chunks per tile
    constexpr int NA = (CA * 8 + 255) / 256, NB = (CB * 8 + 255) / 256;   // 4x4 (pixel x channel) blocks per thread
    __shared__ float lds[2][WK * (BM + BN)];
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
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int64_t kbeg = (int64_t)slice * p.per_split;
    const int64_t kend = (kbeg + p.per_split < p.pixels) ? kbeg + p.per_split : p.pixels;
    const int nks = (int)((kend - kbeg + WK - 1) / WK);
    const int HoWo = p.Ho * p.Wo;

    // Staging.  A thread owns 4x4 blocks: 4 consecutive channels (one 16-byte load per pixel) x 4 consecutive pixels,
    // stored as they come.  Block b of a tile: chunk = b % chunks, pixel group = b / chunks.
    const int ca = tid % CA, pga0 = tid / CA;                // A: block i -> pixel group pga0 + (256 / CA) * i
    const bool a_active = (CA * 8 >= 256) || tid < CA * 8;
    const bool a_col_ok = (m0 + 4 * ca) < p.ldy;
    const int cb = tid % CB, pgb0 = tid / CB;                // B: chunk cb -> fixed (tap, ci0)
    const int jcol = n0 + 4 * cb;
    const int tap = jcol / p.Cin;
    const int ci0 = jcol - tap * p.Cin;
    const int fr = tap / p.kw, fs = tap - fr * p.kw;
    const bool b_col_ok = jcol < p.Kflat;

    // Both operands are read with BUFFER loads, and everything that must read as zero -- image padding, pixels past
    // the end of the K slice, columns past the matrix -- is simply given an out-of-range offset: the hardware range
    // check returns 0.0 for it (checked per dword against num_records, soffset included; tools/probes/buffer_probe.hip).
    // So the load phase has no clamps and no validity masks, the store phase no selects, and for dY not even address
    // arithmetic: four per-thread offsets fixed for the whole kernel + one scalar K-step advance.
    //   dY: descriptor = the K slice [kbeg, kend) only, so "pixel >= kend" is out of range by itself;
    //   X : descriptor starts at the first image the slice touches; the host guarantees the slice's span of images
    //       stays below 2 GiB, so byte offsets are plain int32 and -1 is always out of range.
    const int n_first = (int)(kbeg / HoWo);
    const int rel0 = (int)(kbeg - (int64_t)n_first * HoWo);  // slice-relative pixel index of kbeg within image n_first
    const int64_t img = (int64_t)p.Hi * p.Wi * p.Cin;        // floats per input image
    int64_t xbytes = ((int64_t)p.N - n_first) * img * 4;
    if (xbytes > 0x7FFFFFFF) xbytes = 0x7FFFFFFF;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.dy + kbeg * p.ldy), 0, (unsigned)((kend - kbeg) * p.ldy * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.x + (int64_t)n_first * img), 0, (unsigned)xbytes, 0x00020000);
    unsigned a_voff[NA][4];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int px = 4 * (pga0 + (256 / CA) * i) + q;
            a_voff[i][q] = (a_active && a_col_ok) ? (unsigned)((px * p.ldy + m0 + 4 * ca) * 4) : 0x80000000u;
        }
    const int fr_t = b_col_ok ? fr : (1 << 24);              // a column past the matrix fails every row test below
    const int tap_off = ((fr * p.Wi + fs) * p.Cin + ci0) * 4;

    // Pixel table: one entry per pixel of the K range, (byte offset of input pixel (ih0, iw0) channel 0, ih0, iw0) with
    // ih0 = oh*stride - pad: the two integer divisions of the (n, oh, ow) decomposition are done once per pixel for all
    // 256 threads and all taps.  Filled TB K-steps at a time by the whole workgroup (one entry per thread).
    auto fill_batch = [&](int j) {
        const int rel = rel0 + j * (TB * WK) + tid;           // pixel index relative to image n_first
        int4 e = make_int4(0, -(1 << 28), 0, 0);              // past the slice: fails the row test (also with fr_t = 2^24 added)
        if (kbeg + (int64_t)j * (TB * WK) + tid < kend) {
            const unsigned n = (unsigned)rel / (unsigned)HoWo;
            const unsigned rem = (unsigned)rel - n * (unsigned)HoWo;
            const unsigned oh = rem / (unsigned)p.Wo, ow = rem - oh * (unsigned)p.Wo;
            const int ih0 = (int)oh * p.stride - p.pad, iw0 = (int)ow * p.stride - p.pad;
            e = make_int4((((int)n * p.Hi + ih0) * p.Wi + iw0) * p.Cin * 4, ih0, iw0, 0);
        }
        pixtab[j & 1][tid] = e;
    };
    const bool do_cs = p.colsum != nullptr && (tile % p.tiles_n) == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 va[NA][4], vb[NB][4];
    auto bufload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
    };
    auto load_step = [&](int ks) {
        const unsigned so = (unsigned)ks * (unsigned)(WK * 4) * (unsigned)p.ldy;   // scalar: the K-step advance
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) va[i][q] = bufload(rs_a, a_voff[i][q], so);
        const int4 *tab = &pixtab[(ks / TB) & 1][(ks % TB) * WK];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pg = pgb0 + (256 / CB) * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 e = tab[4 * pg + q];
                const bool ok = (unsigned)(e.y + fr_t) < (unsigned)p.Hi && (unsigned)(e.z + fs) < (unsigned)p.Wi;
                vb[i][q] = bufload(rs_b, ok ? (unsigned)(e.x + tap_off) : 0xFFFFFFFFu, 0u);
            }
        }
    };
    auto store_step = [&](int buf) {
        float *A = lds[buf], *B = lds[buf] + WK * BM;      // [pixel][channel], exactly as the data lies in memory; zeros came from the loads
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!a_active) break;
            const int pg = pga0 + (256 / CA) * i;
            float4 *v = va[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<float4 *>(A + (4 * pg + q) * BM + 4 * ca) = v[q];
                // column sums ride on the staged dY; accumulated here, after the MFMAs, so the loads stay in flight
                if (do_cs) { cs.x += v[q].x; cs.y += v[q].y; cs.z += v[q].z; cs.w += v[q].w; }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pg = pgb0 + (256 / CB) * i;
            float4 *v = vb[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (p.in_relu) { v[q].x = fmaxf(v[q].x, 0.f); v[q].y = fmaxf(v[q].y, 0.f); v[q].z = fmaxf(v[q].z, 0.f); v[q].w = fmaxf(v[q].w, 0.f); }
                *reinterpret_cast<float4 *>(B + (4 * pg + q) * BN + 4 * cb) = v[q];
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    fill_batch(0);
    __syncthreads();
    if (nks > 0) {
        load_step(0);
        store_step(0);
    }
    __syncthreads();
    // Fragments.  v_mfma_f32_32x32x2_f32 wants A[row i = lane&31][k = lane>>5] in one register per lane.  A lane reads TWO
    // adjacent channels of pixel k with one ds_read_b64 and uses them as row i of two different 32-row MFMA tiles: tile tm
    // then holds channels 2*i + tm (a fixed permutation of the wave's 64 channels, undone in the epilogue's index
    // arithmetic).  No transposition anywhere: staging is the plain 16-byte copy, the 32 lanes of a read cover 256
    // contiguous bytes (conflict-free), and every fragment address is base + immediate.
    const int hi = lane >> 5;
    const int fa0 = hi * BM + wm * 64 + 2 * (lane & 31);
    const int fb0 = hi * BN + wn * 64 + 2 * (lane & 31);
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_step(ks + 1);                 // reads table batch (ks+1)/TB, published by an earlier barrier
        const float *A = lds[buf] + fa0;
        const float *B = lds[buf] + WK * BM + fb0;
        float2 fa[3], fb[3];                                  // reads run two k-pairs ahead of the MFMAs that consume them
#pragma unroll
        for (int pre = 0; pre < 2; ++pre) {
            fa[pre] = *reinterpret_cast<const float2 *>(A + 2 * pre * BM);
            fb[pre] = *reinterpret_cast<const float2 *>(B + 2 * pre * BN);
        }
#pragma unroll
        for (int kp = 0; kp < WK / 2; ++kp) {
            const int cur = kp % 3, nxt = (kp + 2) % 3;
            if (kp + 2 < WK / 2) {
                fa[nxt] = *reinterpret_cast<const float2 *>(A + 2 * (kp + 2) * BM);
                fb[nxt] = *reinterpret_cast<const float2 *>(B + 2 * (kp + 2) * BN);
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur].y, acc[1][1], 0, 0, 0);
            // keep that order: (2 LDS reads for k-pair kp+2) then (4 MFMAs of k-pair kp)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        // next table batch: needed from iteration ks = TB*(j+1) - 1 on; its slot held batch j-1, last read at TB*j - 2
        if ((ks % TB) == 3 && (ks / TB + 1) * TB < nks) fill_batch(ks / TB + 1);
        if (ks + 1 < nks) store_step(buf ^ 1);
        __syncthreads();
    }
    if (nks == 0) return;
    if (do_cs) {                                              // the last barrier of the loop freed the staging LDS
        float4 *red = reinterpret_cast<float4 *>(lds[0]);
        constexpr int GA = (CA * 8 >= 256) ? 256 / CA : 8;    // pixel groups that carried A blocks (x NA passes, already summed)
        if (a_active) red[pga0 * CA + ca] = cs;
        __syncthreads();
        if (tid < CA) {
            float4 t = red[tid];
#pragma unroll
            for (int j = 1; j < GA; ++j) { const float4 u = red[j * CA + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            const int c = m0 + 4 * tid;
            if (c + 0 < p.Cout) atomicAdd(p.colsum + c + 0, t.x);
            if (c + 1 < p.Cout) atomicAdd(p.colsum + c + 1, t.y);
            if (c + 2 < p.Cout) atomicAdd(p.colsum + c + 2, t.z);
            if (c + 3 < p.Cout) atomicAdd(p.colsum + c + 3, t.w);
        }
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wn * 64 + 2 * (lane & 31) + tn;          // tile tn holds columns 2*j + tn (see the fragments)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) + tm;   // tile tm: channels 2*i + tm
                if (row < p.Cout && col < p.Kflat) atomicAdd(p.dw + (int64_t)row * p.Kpad + col, acc[tm][tn][e]);
            }
        }
}

extern "C" int rn_conv_wgrad(const float *dy, int ldy, const float *x, float *dw, float *colsum, int N, int Hi, int Wi,
                             int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int in_relu,
                             void *stream) {
    if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin < 4 || (Cin & 3) || (ldy & 3) || ldy < Cout)
        return RN_EINVAL;
    WgradArgs a;
    a.dy = dy; a.x = x; a.dw = dw; a.colsum = colsum; a.ldy = ldy;
    a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad; a.in_relu = in_relu;
    a.Kflat = kh * kw * Cin;
    a.Kpad = (a.Kflat + 31) / 32 * 32;
    a.pixels = (int64_t)N * Ho * Wo;
    const bool narrow_m = Cout <= 64;
    const int BM = narrow_m ? 64 : 128, BN = narrow_m ? 256 : 128;
    const int tiles_m = (Cout + BM - 1) / BM;
    a.tiles_n = (a.Kflat + BN - 1) / BN;
    const int tiles = tiles_m * a.tiles_n;
    // enough K slices to put ~4 workgroups on every CU, each at least 16 K-steps long
    int64_t splits = (1024 + tiles - 1) / tiles;
    const int64_t max_splits = (a.pixels + 16 * WK - 1) / (16 * WK);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    a.xcd_map = tiles >= 16 && splits > 8 && splits + 7 <= max_splits;
    if (a.xcd_map) splits = (splits + 7) / 8 * 8;            // whole slices per XCD: equal shares for the 8
    // Buffer-load addressing (see the kernel): the dY bytes of one K slice and the span of input images a slice can
    // touch must each stay below 2 GiB; more slices make both smaller.
    const int64_t HoWo = (int64_t)Ho * Wo, img_bytes = (int64_t)Hi * Wi * Cin * 4;
    for (;;) {
        a.per_split = ((a.pixels + splits - 1) / splits + WK - 1) / WK * WK;
        const int64_t span_imgs = (a.per_split + HoWo - 2) / HoWo + 1;
        if ((a.per_split + WK) * ldy * 4 <= 0x7FFFFFFF && span_imgs * img_bytes <= 0x7FFFFFFF) break;
        if (a.per_split <= WK || splits >= 65535) return RN_EINVAL;   // a single image of > 2 GiB
        splits = splits * 2 > 65535 ? 65535 : splits * 2;
        a.xcd_map = 0;
    }
    splits = (a.pixels + a.per_split - 1) / a.per_split;
    a.tiles = tiles;
    a.splits = (int)splits;
    const dim3 grid((unsigned)(tiles * (a.xcd_map ? (splits + 7) / 8 * 8 : splits)));
    if (narrow_m)
        hipLaunchKernelGGL((conv_wgrad_kernel<1, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
