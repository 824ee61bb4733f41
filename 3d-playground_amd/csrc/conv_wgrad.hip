// Weight gradient of the convolutions on the fp32 matrix cores.
//
// The reference obtains dW from torch autograd (cuDNN wgrad kernels) for every nn.Conv2d of D/model.py; here it
// is one GEMM per layer:  dW[co][r][s][ci] = sum over output pixels of dY[pixel][co] * X[pixel shifted by the
// tap][ci], i.e.  M = Cout,  N = kh*kw*Cin (the packed-weight row, so a tile may span several taps of a
// narrow layer),  K = N*Ho*Wo pixels.
//
// Both operands are "K-major" in memory already (NHWC: one pixel = one contiguous channel vector), so tiles are
// staged into LDS exactly as they lie: [32 pixels][64*WM channels] and [32 pixels][64*WN flattened (tap,ci)].
// The v_mfma_f32_32x32x2_f32 fragment A[m = lane&31][k = lane>>5] is then one ds_read_b32 with lanes 0-31 on
// consecutive words (conflict-free), one read per MFMA operand tile; 64 reads against 64 MFMAs (64 cycles each)
// per wave per K-step.
//
// K is split over gridDim.y; partial tiles are added with fp32 atomics (two 128-byte row segments per wave
// instruction, the full-rate shape) into a buffer the caller zeroes -- the five pyramid levels of a shared head
// accumulate into the same buffer.  Blocks of one K-slice are adjacent in the grid so they stream the same dY /
// X slabs through one XCD's L2.
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
    int64_t pixels, per_split;   // K extent and K per grid.y slice (multiple of WK)
};

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int CA = BM / 4, CB = BN / 4;                  // 16-byte chunks per staged row
    constexpr int PA = CA / 8, PB = CB / 8;                  // passes (rows per pass = 256 / chunks)
    constexpr int RA = 256 / CA, RB = 256 / CB;
    __shared__ float lds[2][WK * (BM + BN)];
    __shared__ int4 pixtab[2][WK];                           // per K-step: (n*Hi, oh*st - pad, ow*st - pad, valid)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = (blockIdx.x / p.tiles_n) * BM, n0 = (blockIdx.x % p.tiles_n) * BN;
    const int64_t kbeg = (int64_t)blockIdx.y * p.per_split;
    const int64_t kend = (kbeg + p.per_split < p.pixels) ? kbeg + p.per_split : p.pixels;
    const int nks = (int)((kend - kbeg + WK - 1) / WK);
    const int HoWo = p.Ho * p.Wo;

    // A staging: chunk ca of the co range, rows (tid / CA) + RA*i
    const int ca = tid % CA, ra0 = tid / CA;
    const bool a_col_ok = (m0 + 4 * ca) < p.ldy;
    // B staging: chunk cb -> fixed (tap, ci0)
    const int cb = tid % CB, rb0 = tid / CB;
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
    const bool do_cs = p.colsum != nullptr && (blockIdx.x % p.tiles_n) == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 va[PA], vb[PB];
    auto load_step = [&](int ks) {
        const int64_t kb = kbeg + (int64_t)ks * WK;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int64_t pix = kb + ra0 + RA * i;
            va[i] = (a_col_ok && pix < kend) ? *reinterpret_cast<const float4 *>(p.dy + pix * p.ldy + m0 + 4 * ca)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int4 e = pixtab[ks & 1][rb0 + RB * i];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ih = e.y + fr, iw = e.z + fs;
            if (b_col_ok && e.w && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi) {
                v = *reinterpret_cast<const float4 *>(p.x + ((int64_t)(e.x + ih) * p.Wi + iw) * p.Cin + ci0);
                if (p.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            vb[i] = v;
        }
    };
    auto store_step = [&](int buf) {
        float *A = lds[buf], *B = lds[buf] + WK * BM;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            *reinterpret_cast<float4 *>(A + (ra0 + RA * i) * BM + 4 * ca) = va[i];
            // column sums ride on the staged dY; accumulated here, after the MFMAs, so the loads stay in flight
            if (do_cs) { cs.x += va[i].x; cs.y += va[i].y; cs.z += va[i].z; cs.w += va[i].w; }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<float4 *>(B + (rb0 + RB * i) * BN + 4 * cb) = vb[i];
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
    const int fa = (lane >> 5) * BM + wm * 64 + (lane & 31);
    const int fb = (lane >> 5) * BN + wn * 64 + (lane & 31);
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_step(ks + 1);                 // reads pixtab[(ks+1)&1], published by an earlier barrier
        const float *A = lds[buf] + fa;
        const float *B = lds[buf] + WK * BM + fb;
        // fragment reads run two k-pairs ahead of the MFMAs that consume them (distinct registers per slot), so an
        // MFMA group never waits on a read issued one instruction earlier
        float fa[3][2], fb[3][2];
#pragma unroll
        for (int pre = 0; pre < 2; ++pre) {
            fa[pre][0] = A[2 * pre * BM]; fa[pre][1] = A[2 * pre * BM + 32];
            fb[pre][0] = B[2 * pre * BN]; fb[pre][1] = B[2 * pre * BN + 32];
        }
#pragma unroll
        for (int kp = 0; kp < WK / 2; ++kp) {
            const int cur = kp % 3, nxt = (kp + 2) % 3;
            if (kp + 2 < WK / 2) {
                fa[nxt][0] = A[2 * (kp + 2) * BM]; fa[nxt][1] = A[2 * (kp + 2) * BM + 32];
                fb[nxt][0] = B[2 * (kp + 2) * BN]; fb[nxt][1] = B[2 * (kp + 2) * BN + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
            // keep that order: (2 LDS reads for k-pair kp+2) then (4 MFMAs of k-pair kp)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (ks + 2 < nks) fill_table(ks + 2);                 // slot [ks&1] was last read by load_step(ks), before the previous barrier
        if (ks + 1 < nks) store_step(buf ^ 1);
        __syncthreads();
    }
    if (nks == 0) return;
    if (do_cs) {                                              // the last barrier of the loop freed the staging LDS
        float4 *red = reinterpret_cast<float4 *>(lds[0]);
        red[ra0 * CA + ca] = cs;
        __syncthreads();
        if (ra0 == 0) {
            float4 t = red[ca];
#pragma unroll
            for (int j = 1; j < RA; ++j) { const float4 u = red[j * CA + ca]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            const int c = m0 + 4 * ca;
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
    a.per_split = ((a.pixels + splits - 1) / splits + WK - 1) / WK * WK;
    splits = (a.pixels + a.per_split - 1) / a.per_split;
    const dim3 grid(tiles, (unsigned)splits);
    if (narrow_m)
        hipLaunchKernelGGL((conv_wgrad_kernel<1, 4>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
