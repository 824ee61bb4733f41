// Weight gradient of the convolutions on the fp32 matrix cores.
//
// The reference obtains dW from torch autograd (cuDNN wgrad kernels) for every nn.Conv2d of D/model.py; here it
// is one GEMM per layer:  dW[co][r][s][ci] = sum over output pixels of dY[pixel][co] * X[pixel shifted by the
// tap][ci], i.e.  M = Cout,  N = kh*kw*Cin (the packed-weight row, so a tile may span several taps of a
// narrow layer),  K = N*Ho*Wo pixels.
//
// Both operands are pixel-major in memory (NHWC: one pixel = one contiguous channel vector) while the MFMA wants
// each lane's k operands -- pixels -- of one channel row.  A thread therefore stages 4x4 blocks (4 pixels x 4
// channels: four 16-byte global loads), transposes them in registers for free and writes four 16-byte LDS rows into
// a [channel][32 pixels] tile (XOR-swizzled, conflict-free both ways); a fragment is then one ds_read_b128 = the k
// operands of 4 MFMAs: 16 LDS reads against 64 MFMAs (64 cycles each) per wave per K-step.
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

// LDS tile layout: [channel row][32 pixels], pixel-contiguous, no padding; the 16-byte slot (4 pixels) of a row is
// XOR-swizzled with s(row) = (row & 7) ^ ((row >> 3) & 7).  Both access patterns then touch 8 distinct slots per 8
// consecutive lanes: the staging writes (lane -> rows 4*ca + j, fixed slot) and the fragment reads (lane -> 32
// consecutive rows, fixed slot).
__device__ __forceinline__ int wg_swz(int row) { return (row & 7) ^ ((row >> 3) & 7); }

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int CA = BM / 4, CB = BN / 4;                  // 4-channel chunks per tile
    constexpr int NA = (CA * 8 + 255) / 256, NB = (CB * 8 + 255) / 256;   // 4x4 (pixel x channel) blocks per thread
    __shared__ float lds[2][WK * (BM + BN)];
    __shared__ int4 pixtab[2][WK];                           // per K-step: (n*Hi, oh*st - pad, ow*st - pad, valid)

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

    // Staging.  A thread owns 4x4 blocks: 4 consecutive channels (one 16-byte global load per pixel) x 4 consecutive
    // pixels; the block is transposed in registers (free: it is only a renaming) and leaves as four 16-byte LDS
    // writes, one per channel row, 4 pixels each.  Block b of a tile: chunk = b % chunks, pixel group = b / chunks.
    const int ca = tid % CA, pga0 = tid / CA;                // A: block i -> pixel group pga0 + (256 / CA) * i
    const bool a_active = (CA * 8 >= 256) || tid < CA * 8;
    const bool a_col_ok = (m0 + 4 * ca) < p.ldy;
    const int cb = tid % CB, pgb0 = tid / CB;                // B: chunk cb -> fixed (tap, ci0)
    const int jcol = n0 + 4 * cb;
    const int tap = jcol / p.Cin;
    const int ci0 = jcol - tap * p.Cin;
    const int fr = tap / p.kw, fs = tap - fr * p.kw;
    const bool b_col_ok = jcol < p.Kflat;

    // One lane per pixel of a K-step decomposes it into (n, oh, ow) for everybody: two integer divisions per
    // K-step instead of two per staged row.
    auto fill_table = [&](int ks) {
        if (tid < WK) {
            const int64_t pix = kbeg + (int64_t)ks * WK + tid;
            int4 e = make_int4(0, 0, 0, 0);
            if (pix < kend) {
                const int n = (int)(pix / HoWo);
                const int rem = (int)(pix - (int64_t)n * HoWo);
                const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
                e = make_int4(n * p.Hi, oh * p.stride - p.pad, ow * p.stride - p.pad, 1);
            }
            pixtab[ks & 1][tid] = e;
        }
    };
    const bool do_cs = p.colsum != nullptr && (tile % p.tiles_n) == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 va[NA][4], vb[NB][4];
    unsigned okmask = 0;                                     // validity of the staged blocks: bit i*4+q for A, 16 + i*4+q for B
    // Loads are unconditional, from clamped (always mapped) addresses; what must read as zero (image padding, rows
    // past the end of the K range, columns past the tile) is zeroed in store_step, after the MFMAs.  A select or a
    // branch right behind a load would make the compiler wait for it here, one round trip per load.
    const int64_t a_col = a_col_ok ? (int64_t)(m0 + 4 * ca) : 0;
    const int ci_safe = b_col_ok ? ci0 : 0;
    const int fr_safe = b_col_ok ? fr : 0, fs_safe = b_col_ok ? fs : 0;
    auto load_step = [&](int ks) {
        const int64_t kb = kbeg + (int64_t)ks * WK;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pg = pga0 + (256 / CA) * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t pix = kb + 4 * pg + q;
                const bool ok = a_active && a_col_ok && pix < kend;
                okmask |= ok ? (1u << (i * 4 + q)) : 0u;
                const int64_t pc = pix < p.pixels ? pix : p.pixels - 1;
                va[i][q] = *reinterpret_cast<const float4 *>(p.dy + pc * p.ldy + a_col);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pg = pgb0 + (256 / CB) * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 e = pixtab[ks & 1][4 * pg + q];
                const int ih = e.y + fr_safe, iw = e.z + fs_safe;
                const bool ok = b_col_ok && e.w && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
                okmask |= ok ? (1u << (16 + i * 4 + q)) : 0u;
                const int ihc = ih < 0 ? 0 : (ih >= p.Hi ? p.Hi - 1 : ih);
                const int iwc = iw < 0 ? 0 : (iw >= p.Wi ? p.Wi - 1 : iw);
                vb[i][q] = *reinterpret_cast<const float4 *>(p.x + ((int64_t)(e.x + ihc) * p.Wi + iwc) * p.Cin + ci_safe);
            }
        }
    };
    auto store_step = [&](int buf) {
        float *A = lds[buf], *B = lds[buf] + WK * BM;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!a_active) break;
            const int pg = pga0 + (256 / CA) * i;
            float4 *v = va[i];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (!(okmask & (1u << (i * 4 + q)))) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int r = 4 * ca;                                        // rows r..r+3 share (r >> 3) only pairwise: swizzle per row
            *reinterpret_cast<float4 *>(A + (r + 0) * WK + 4 * (pg ^ wg_swz(r + 0))) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
            *reinterpret_cast<float4 *>(A + (r + 1) * WK + 4 * (pg ^ wg_swz(r + 1))) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
            *reinterpret_cast<float4 *>(A + (r + 2) * WK + 4 * (pg ^ wg_swz(r + 2))) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
            *reinterpret_cast<float4 *>(A + (r + 3) * WK + 4 * (pg ^ wg_swz(r + 3))) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
            // column sums ride on the staged dY; accumulated here, after the MFMAs, so the loads stay in flight
            if (do_cs) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { cs.x += v[q].x; cs.y += v[q].y; cs.z += v[q].z; cs.w += v[q].w; }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pg = pgb0 + (256 / CB) * i;
            float4 *v = vb[i];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (!(okmask & (1u << (16 + i * 4 + q)))) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.in_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q].x = fmaxf(v[q].x, 0.f); v[q].y = fmaxf(v[q].y, 0.f); v[q].z = fmaxf(v[q].z, 0.f); v[q].w = fmaxf(v[q].w, 0.f); }
            }
            const int r = 4 * cb;
            *reinterpret_cast<float4 *>(B + (r + 0) * WK + 4 * (pg ^ wg_swz(r + 0))) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
            *reinterpret_cast<float4 *>(B + (r + 1) * WK + 4 * (pg ^ wg_swz(r + 1))) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
            *reinterpret_cast<float4 *>(B + (r + 2) * WK + 4 * (pg ^ wg_swz(r + 2))) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
            *reinterpret_cast<float4 *>(B + (r + 3) * WK + 4 * (pg ^ wg_swz(r + 3))) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    fill_table(0);
    fill_table(1);
    __syncthreads();
    if (nks > 0) {
        load_step(0);
        store_step(0);
    }
    __syncthreads();
    // fragment rows of this lane: A tiles tm = 0,1 and B tiles tn = 0,1; v_mfma_f32_32x32x2_f32 wants A[m = lane&31][k = lane>>5]:
    // one ds_read_b128 gives 4 consecutive pixels of the row = the k operands of 4 MFMAs (lanes 0-31 take pixel group 2t,
    // lanes 32-63 group 2t+1; A and B use the same assignment, so the products pair up the same pixels)
    const int hi = lane >> 5;
    int rowA[2], rowB[2], swA[2], swB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        rowA[t] = wm * 64 + t * 32 + (lane & 31);
        rowB[t] = wn * 64 + t * 32 + (lane & 31);
        swA[t] = wg_swz(rowA[t]);
        swB[t] = wg_swz(rowB[t]);
    }
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_step(ks + 1);                 // reads pixtab[(ks+1)&1], published by an earlier barrier
        const float *A = lds[buf];
        const float *B = lds[buf] + WK * BM;
        float4 fa[2][2], fb[2][2];                            // [slot][tile]: reads run one group of 16 MFMAs ahead
        auto read_frag = [&](int t, float4 (&a)[2], float4 (&b)[2]) {
            const int kg = 2 * t + hi;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                a[q] = *reinterpret_cast<const float4 *>(A + rowA[q] * WK + 4 * (kg ^ swA[q]));
                b[q] = *reinterpret_cast<const float4 *>(B + rowB[q] * WK + 4 * (kg ^ swB[q]));
            }
        };
        read_frag(0, fa[0], fb[0]);
#pragma unroll
        for (int t = 0; t < WK / 8; ++t) {
            const int cur = t & 1;
            if (t + 1 < WK / 8) read_frag(t + 1, fa[cur ^ 1], fb[cur ^ 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a0 = j == 0 ? fa[cur][0].x : j == 1 ? fa[cur][0].y : j == 2 ? fa[cur][0].z : fa[cur][0].w;
                const float a1 = j == 0 ? fa[cur][1].x : j == 1 ? fa[cur][1].y : j == 2 ? fa[cur][1].z : fa[cur][1].w;
                const float b0 = j == 0 ? fb[cur][0].x : j == 1 ? fb[cur][0].y : j == 2 ? fb[cur][0].z : fb[cur][0].w;
                const float b1 = j == 0 ? fb[cur][1].x : j == 1 ? fb[cur][1].y : j == 2 ? fb[cur][1].z : fb[cur][1].w;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (ks + 2 < nks) fill_table(ks + 2);                 // slot [ks&1] was last read by load_step(ks), before the previous barrier
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
            const int col = n0 + wn * 64 + tn * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
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
    a.per_split = ((a.pixels + splits - 1) / splits + WK - 1) / WK * WK;
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
