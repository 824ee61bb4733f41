// Winograd F(4x4, 3x3) for the 3x3 / stride 1 / padding 1 convolutions of the head towers (RegressionModel /
// ClassificationModel conv1..conv4, D/model.py:120-205): a quarter of the multiplications of the direct form.
//
//     y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        d: 6x6 input patch, g: 3x3 filter, y: 4x4 outputs
//
// Three stages, all of them streaming except the middle one:
//   wino_in_kernel    x [N,H,W,C]          -> V [36][T][C]      B^T d B per tile and channel        (HBM-bound)
//   conv_igemm        V_p [T][C] x U_p     -> M [36][T][Cout]   36 independent GEMMs, ONE launch of the implicit-GEMM
//                                                               kernel as a 1x1 convolution over 36 "images" with
//                                                               per-image weights (rn_conv_desc.w_batch_stride)
//   wino_out_kernel   M [36][T][Cout]      -> y [N,H,W,Cout]    A^T m A + the conv epilogue          (HBM-bound)
// T = tiles of all problems of a group (the pyramid levels), padded to a multiple of 256 so that no GEMM tile straddles
// two of the 36 positions.  The same three stages compute the data gradient: dY goes through wino_in_kernel, the weights
// are transformed from the flipped, transposed filter (rn_wino_weights mode 1), and the output stage applies the ReLU
// mask / gradient accumulation of the dgrad epilogue.
//
// Numerics: fp32 throughout; the transform matrices of F(4x4,3x3) (entries up to 8 and down to 1/24) cost about one
// decimal digit: 8e-6 of the output's max magnitude against 3e-7 for the direct kernel (measured against fp64).  That
// is inside the 1e-4 contract but not free, so the path is used where it pays most (256-channel towers, training) and
// the direct kernel stays the default everywhere else.
//
// A thread owns one tile x 4 channels: 36 16-byte loads (a wave reads one pixel's whole channel vector per load: 1 KiB
// contiguous), the two 1-D transforms in registers, 36 (16) 16-byte stores.
#include "common.h"
#include "mfma_split.h"

__device__ __forceinline__ float4 f4(float s) { return make_float4(s, s, s, s); }
// M (the GEMM's result: written once, read once by the output transform, larger than the caches) is read with the nontemporal
// hint: wino_out 7.51 -> 6.98 ms per training step.  The same hint on the STORES of V / Z made the input transforms slower
// (8.6 -> 9.3 ms): they stay plain.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float *p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

// o = B^T d for one 6-vector (B^T of Lavin & Gray, F(4x4,3x3))
__device__ __forceinline__ void bt6(const float4 d[6], float4 o[6]) {
    o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    o[1] = d[3] + d[4] - 4.f * (d[1] + d[2]);
    o[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    o[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    o[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// o = A^T m for one 6-vector
__device__ __forceinline__ void at6(const float4 m[6], float4 o[4]) {
    const float4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m[5];
}


// Grouped launches: up to RN_MAX_GROUP problems (pyramid levels) whose tiles are concatenated in V / M / Z in table
// order, so a global tile index is both the row of the transformed tensor and, through the table, a (problem, n, th, tw).
struct WinoProblem { int N, H, W, TH, TW; };
struct WinoTable {
    int n;
    int64_t tile_end[RN_MAX_GROUP];              // exclusive prefix sums of the problems' tile counts
    WinoProblem p[RN_MAX_GROUP];
    const float *src[RN_MAX_GROUP];              // x / dy (input transforms), unused by the output transform
    float *dst[RN_MAX_GROUP];                    // y (output transform)
    const float *add[RN_MAX_GROUP];
    const float *mask[RN_MAX_GROUP];
    unsigned *sign[RN_MAX_GROUP];                // output transform: sign bits of y (common.h: rn_sign_store), or NULL
    unsigned *amax[RN_MAX_GROUP];                // amax tables, one per image (mfma_split.h; RN_FP32_SPLIT3), or NULL: of src (input
                                                 // transforms, read) / of dst (output transform, written)
};

// Amax of a Winograd-domain tensor (RN_FP32_SPLIT3), as plain WORDS (an fp32 bit pattern whose exponent field is the bound).  The transforms are linear maps of bounded gain -- |B^T d B| <= 100 max|d|
// (< 2^7), |A dy A^T| <= 225 max|dy| (< 2^8): the absolute row sums of B^T are at most 10, of A at most 15 -- so the word of a tile's row
// of V / Z follows from the largest exponent in its IMAGE's table of the untransformed tensor by adding the gain: no reduction, no atomics, and
// an image's scales still depend on that image alone.  The GEMM of the forward / data gradient takes the ROW words (rn_conv_desc.x_amax
// with strides (0, 1)); the weight gradient, which reduces over the tiles of all images, takes the one TENSOR word (the largest of all).
#define WINO_GAIN_BT 7
#define WINO_GAIN_A 8
__device__ __forceinline__ void wino_tensor_word(const WinoTable &g, unsigned *tensor_amax, int gain) {
    if (tensor_amax == nullptr || blockIdx.x != 0 || threadIdx.x >= 64) return;      // wave 0 of workgroup 0, all 64 lanes
    int e = 0;
#pragma unroll
    for (int i = 0; i < RN_MAX_GROUP; ++i)
        if (i < g.n && g.amax[i] != nullptr) { const int ei = rn_amax_exp_all(g.amax[i], g.p[i].N); e = ei > e ? ei : e; }
    if (threadIdx.x == 0) atomicMax(tensor_amax, rn_amax_word_of_exp(e, gain));     // (several launches may fill one tensor: the caller zeroes the word)
}
__device__ __forceinline__ int wino_locate(const WinoTable &g, int64_t gt, int64_t &local) {
    int q = 0;
#pragma unroll
    for (int i = 0; i < RN_MAX_GROUP - 1; ++i) q += (i + 1 < g.n && gt >= g.tile_end[i]) ? 1 : 0;
    int64_t first = 0;
#pragma unroll
    for (int i = 1; i < RN_MAX_GROUP; ++i)
        if (q == i) first = g.tile_end[i - 1];
    local = gt - first;
    return q;
}

// ---------------------------------------------------------------------------------------------- input transform
#define WINO_SELECT(q, g, pr, EXTRA)                          \
    WinoProblem pr = g.p[0];                                   \
    _Pragma("unroll") for (int i_ = 1; i_ < RN_MAX_GROUP; ++i_) \
        if (q == i_) { pr = g.p[i_]; EXTRA }

__global__ __launch_bounds__(256) void wino_in_kernel(const WinoTable g, float *__restrict__ V, int C, int64_t t0, int64_t Tpad,
                                                      unsigned *__restrict__ row_amax, unsigned *__restrict__ tensor_amax) {
    wino_tensor_word(g, tensor_amax, WINO_GAIN_BT);
    const int cq = C >> 2;
    // 6x6 patches of neighbouring tiles overlap by two pixels: every input pixel is read by 2.25 tiles.  With the plain id the
    // tiles of one image row are spread over all eight XCDs and each L2 fetched its own copy of the shared pixels (PMC: 2.06x
    // the input from HBM); one contiguous band of tiles per XCD keeps the re-reads in that L2.
    const int64_t id = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int64_t gt_raw = id / cq, t_last = g.tile_end[g.n - 1];
    const int64_t gt = gt_raw < t_last ? gt_raw : t_last - 1;   // (threads past the end still help with the table read below)
    const int c4 = (int)(id - gt_raw * cq) * 4;
    int64_t tile;
    const int q = wino_locate(g, gt, tile);
    const float *x = g.src[0];
    const unsigned *xw = g.amax[0];
    WINO_SELECT(q, g, pr, x = g.src[i_]; xw = g.amax[i_];)
    const int H = pr.H, W = pr.W, TW = pr.TW;
    const int n = (int)(tile / (pr.TH * TW));
    const int r = (int)(tile - (int64_t)n * pr.TH * TW);
    const int th = r / TW, tw = r - th * TW;
    const int h0 = 4 * th - 1, w0 = 4 * tw - 1;
    const float *xb = x + (int64_t)n * H * W * C + c4;
    if (row_amax != nullptr) {                                // wave-cooperative: every lane the largest exponent of ITS image's table
        const int e_img = rn_amax_exp_lanes(xw != nullptr ? reinterpret_cast<const unsigned char *>(xw) + (int64_t)n * RN_AMAX_BYTES : nullptr);
        if (gt_raw < t_last && c4 == 0) row_amax[t0 + gt] = rn_amax_word_of_exp(e_img, WINO_GAIN_BT);
    }
    if (gt_raw >= t_last) return;
    float4 t[6][6];                                           // t[i][j] = (B^T d)[i][j]: columns first
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float4 d[6], o[6];
        const int w = w0 + j;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int h = h0 + i;
#ifdef RN_WINO_ABL       // knock-out (timing only, wrong results): no halo loads.  Measured on the head-tower group: 0.249 -> 0.227 ms,
                         // i.e. the 2.25x re-reads through L2 cost 9 %; the rest is the HBM stream itself (1.17 GB at 5.1 TB/s)
            const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W && i >= 1 && i <= 4 && j >= 1 && j <= 4;
#else
            const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
#endif
            d[i] = ok ? *reinterpret_cast<const float4 *>(xb + ((int64_t)h * W + w) * C) : f4(0.f);
        }
        bt6(d, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i][j] = o[i];
    }
    float *vb = V + (t0 + gt) * C + c4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {                              // rows: V[i][.] = t[i][.] B
        float4 o[6];
        bt6(t[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) st_stream(vb + (int64_t)(i * 6 + j) * Tpad * C, o[j]);
    }
}

// ---------------------------------------------------------------------------------------------- output transform
// mask_mode / add / act as in rn_conv_desc (mask and add have the geometry of y, dense [N,H,W,Cout]).
template <bool PBITS, bool PADD>
__global__ __launch_bounds__(256, 2) void wino_out_kernel(const WinoTable g, const float *__restrict__ M, int Cout, int64_t t0,
                                                       int64_t Tpad, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, int mask_mode, int act, int64_t y_bs) {
    const int cq = Cout >> 2;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t gt = id / cq;
    // (the body sits in a one-trip loop so that threads past the last tile leave it by `break` and reach the amax commit at the end)
    float am = 0.f;                                           // largest |y| this thread stores (the result's amax word of image n)
    unsigned *yam = nullptr;
    int n_img = 0;
    bool live = false;
    do {
    if (gt >= g.tile_end[g.n - 1]) break;
    live = true;
    const int c4 = (int)(id - gt * cq) * 4;
    int64_t tile;
    const int q = wino_locate(g, gt, tile);
    float *y = g.dst[0];
    const float *add = g.add[0], *mask = g.mask[0];
    unsigned *sign = g.sign[0];
    yam = g.amax[0];
    WINO_SELECT(q, g, pr, y = g.dst[i_]; add = g.add[i_]; mask = g.mask[i_]; sign = g.sign[i_]; yam = g.amax[i_];)
    n_img = (int)(tile / (pr.TH * pr.TW));
    const bool mbits = (mask_mode & RN_MASK_BITS) != 0;
    mask_mode &= 3;
    const int H = pr.H, W = pr.W, TW = pr.TW;
    const int n = (int)(tile / (pr.TH * TW));
    const int r = (int)(tile - (int64_t)n * pr.TH * TW);
    const int th = r / TW, tw = r - th * TW;
    const float *mb = M + (t0 + gt) * Cout + c4;
    float4 t[4][6];                                           // t[i][j] = (A^T m)[i][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float4 m[6], o[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) m[i] = ld_stream(mb + (int64_t)(i * 6 + j) * Tpad * Cout);
        at6(m, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i][j] = o[i];
    }
    float4 sc = f4(1.f), sh = f4(0.f);
    if (scale) sc = *reinterpret_cast<const float4 *>(scale + c4);
    if (shift) sh = *reinterpret_cast<const float4 *>(shift + c4);
    const int64_t ybs = y_bs ? y_bs : (int64_t)H * W * Cout;
    // PBITS: the launch reads ReLU-mask bits; PADD: every problem of the launch has an addend (the launcher's choice of instance).  Their
    // operands are requested a ROW of four pixels at a time, in one batch: loaded pixel by pixel every load sat between two stores with an
    // s_waitcnt vmcnt(0) of its own -- sixteen (with an addend: thirty-two) memory round trips one after the other per thread, and the
    // data-gradient form ran 12 % (with an addend: 68 %) behind the plain one (tools/dbg/wino_out_variants.py).  All sixteen at once
    // need 280-340 registers (one wave per SIMD) or spill; a row costs 4 + 16.  Pixels past the edge read the edge pixel's operands.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 o[4];
        at6(t[i], o);
        const int oh = 4 * th + i;
        if (oh >= H) break;
        unsigned mkw[4];
        float4 adv[4];
        if constexpr (PBITS || PADD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int owc = 4 * tw + j < W ? 4 * tw + j : W - 1;
                const int64_t off = (((int64_t)n * H + oh) * W + owc) * Cout + c4;
                if constexpr (PBITS) mkw[j] = reinterpret_cast<const unsigned *>(mask)[off >> 5] >> ((unsigned)off & 28u);
                if constexpr (PADD) adv[j] = *reinterpret_cast<const float4 *>(add + off);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ow = 4 * tw + j;
            if (ow >= W) break;
            const int64_t off = (((int64_t)n * H + oh) * W + ow) * Cout + c4;      // mask / add: dense
            const int64_t yoff = (int64_t)n * ybs + ((int64_t)oh * W + ow) * Cout + c4;   // y: batch stride (a slice of [B, A, n])
            float v[4] = {sc.x * o[j].x + sh.x, sc.y * o[j].y + sh.y, sc.z * o[j].z + sh.z, sc.w * o[j].w + sh.w};
            float mk[4] = {1.f, 1.f, 1.f, 1.f}, ad[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (PBITS) {
                const unsigned nb = mkw[j];
                mk[0] = (nb & 1u) ? 1.f : 0.f; mk[1] = (nb & 2u) ? 1.f : 0.f; mk[2] = (nb & 4u) ? 1.f : 0.f; mk[3] = (nb & 8u) ? 1.f : 0.f;
            } else if (mask_mode != 0) { const float4 q4 = rn_mask_load4(mask, off, mbits); mk[0] = q4.x; mk[1] = q4.y; mk[2] = q4.z; mk[3] = q4.w; }
            if constexpr (PADD) { const float4 q4 = adv[j]; ad[0] = q4.x; ad[1] = q4.y; ad[2] = q4.z; ad[3] = q4.w; }
            else if (add) { const float4 q4 = *reinterpret_cast<const float4 *>(add + off); ad[0] = q4.x; ad[1] = q4.y; ad[2] = q4.z; ad[3] = q4.w; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float u = v[k];
                if (mask_mode == 1) u = mk[k] > 0.f ? u : 0.f;
                u += ad[k];
                if (act == 1) u = fmaxf(u, 0.f);
                else if (act == 2) u = 1.0f / (1.0f + expf(-u));
                if (mask_mode == 2) u = mk[k] > 0.f ? u : 0.f;
                v[k] = u;
            }
            *reinterpret_cast<float4 *>(y + yoff) = make_float4(v[0], v[1], v[2], v[3]);
            if (sign != nullptr) rn_sign_store(sign, off, v[0], v[1], v[2], v[3]);      // (dense geometry: the 8 lanes of a word share the pixel)
            am = fmaxf(fmaxf(am, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    }
    } while (0);
    if (live) rn_amax_note(yam, n_img, am);
}

// ---------------------------------------------------------------------------------------------- weight transform
// U[p = 6i + j][row][k] = (G g G^T)[i][j];  mode 0 (forward): row = co, k = ci, g = w[co][ci];
// mode 1 (data gradient): row = ci, k = co, g = w[co][ci] rotated by 180 degrees, times scale[co] (folded batch norm).
__global__ void wino_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int Cout, int Cin, int mode,
                                   const float *__restrict__ scale, int rows, int Kpad) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)rows * Kpad) return;
    const int row = (int)(id / Kpad), k = (int)(id - (int64_t)row * Kpad);
    const int kdim = mode == 0 ? Cin : Cout;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float v = 0.f;
            if (k < kdim) {
                if (mode == 0) v = w[(((int64_t)row * Cin + k) * 3 + r) * 3 + s];
                else v = w[(((int64_t)k * Cin + row) * 3 + (2 - r)) * 3 + (2 - s)] * (scale ? scale[k] : 1.f);
            }
            g[r][s] = v;
        }
    // G g: 6x3, then (G g) G^T: 6x6.  G rows: [1/4,0,0] [-1/6,-1/6,-1/6] [-1/6,1/6,-1/6] [1/24,1/12,1/6] [1/24,-1/12,1/6] [0,0,1]
    float t[6][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float a = g[0][s], b = g[1][s], c = g[2][s];
        t[0][s] = a * (1.f / 4.f);
        t[1][s] = -(a + b + c) * (1.f / 6.f);
        t[2][s] = (b - a - c) * (1.f / 6.f);
        t[3][s] = a * (1.f / 24.f) + b * (1.f / 12.f) + c * (1.f / 6.f);
        t[4][s] = a * (1.f / 24.f) - b * (1.f / 12.f) + c * (1.f / 6.f);
        t[5][s] = c;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float a = t[i][0], b = t[i][1], c = t[i][2];
        const float o[6] = {a * (1.f / 4.f), -(a + b + c) * (1.f / 6.f), (b - a - c) * (1.f / 6.f),
                            a * (1.f / 24.f) + b * (1.f / 12.f) + c * (1.f / 6.f),
                            a * (1.f / 24.f) - b * (1.f / 12.f) + c * (1.f / 6.f), c};
#pragma unroll
        for (int j = 0; j < 6; ++j) U[((int64_t)(i * 6 + j) * rows + row) * Kpad + k] = o[j];
    }
}

// ---------------------------------------------------------------------------------------------- weight gradient
// dL/dg = G^T [ sum over tiles of (A dy A^T) .* (B^T d B) ] G.  Z = A dy A^T (4x4 output-gradient tile -> 6x6) is this
// kernel; V = B^T d B is wino_in_kernel; dU[p] = Z_p^T V_p is the weight-gradient GEMM kernel run as 36 batched 1x1
// problems (rn_conv_wgrad_batched); wino_dw_kernel folds G^T dU G into the layer's packed gradient.  Z at position
// (1,1) is the plain sum of the tile's dy, so the bias gradient is the column sum of that one batch entry.
// o = A v for one 4-vector (A = transpose of the A^T of at6)
__device__ __forceinline__ void a6(const float4 v[4], float4 o[6]) {
    const float4 s02 = v[0] + v[2], s13 = v[1] + v[3];
    o[0] = v[0];
    o[1] = s02 + s13;
    o[2] = s02 - s13;
    const float4 e = v[0] + 4.f * v[2], f = 2.f * v[1] + 8.f * v[3];
    o[3] = e + f;
    o[4] = e - f;
    o[5] = v[3];
}

__global__ __launch_bounds__(256) void wino_dy_kernel(const WinoTable g, float *__restrict__ Z, int C, int64_t t0, int64_t Tpad,
                                                      unsigned *__restrict__ tensor_amax) {
    wino_tensor_word(g, tensor_amax, WINO_GAIN_A);
    const int cq = C >> 2;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t gt = id / cq;
    if (gt >= g.tile_end[g.n - 1]) return;
    const int c4 = (int)(id - gt * cq) * 4;
    int64_t tile;
    const int q = wino_locate(g, gt, tile);
    const float *dy = g.src[0];
    WINO_SELECT(q, g, pr, dy = g.src[i_];)
    const int H = pr.H, W = pr.W, TW = pr.TW;
    const int n = (int)(tile / (pr.TH * TW));
    const int r = (int)(tile - (int64_t)n * pr.TH * TW);
    const int th = r / TW, tw = r - th * TW;
    const float *yb = dy + (int64_t)n * H * W * C + c4;
    float4 t[6][4];                                           // t[a][j] = (A dy)[a][j]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 v[4], o[6];
        const int w = 4 * tw + j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int h = 4 * th + i;
            v[i] = (h < H && w < W) ? *reinterpret_cast<const float4 *>(yb + ((int64_t)h * W + w) * C) : f4(0.f);
        }
        a6(v, o);
#pragma unroll
        for (int a = 0; a < 6; ++a) t[a][j] = o[a];
    }
    float *zb = Z + (t0 + gt) * C + c4;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float4 o[6];
        a6(t[a], o);
#pragma unroll
        for (int b = 0; b < 6; ++b) st_stream(zb + (int64_t)(a * 6 + b) * Tpad * C, o[b]);
    }
}

// Both transforms of an output gradient in ONE pass over it: V = B^T dy B (the data gradient's input transform: 6x6 patches with
// their halo) and Z = A dy A^T (the weight gradient's: the patch's inner 4x4).  The backward of a Winograd layer needs both and
// read dy twice; the patch a thread has loaded for V holds the pixels of Z.
__global__ __launch_bounds__(256) void wino_in_both_kernel(const WinoTable g, float *__restrict__ V, float *__restrict__ Z, int C,
                                                           int64_t t0, int64_t Tpad, unsigned *__restrict__ v_row_amax,
                                                           unsigned *__restrict__ z_tensor_amax) {
    wino_tensor_word(g, z_tensor_amax, WINO_GAIN_A);
    const int cq = C >> 2;
    const int64_t id = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int64_t gt_raw = id / cq, t_last = g.tile_end[g.n - 1];
    const int64_t gt = gt_raw < t_last ? gt_raw : t_last - 1;   // (threads past the end still help with the table read below)
    const int c4 = (int)(id - gt_raw * cq) * 4;
    int64_t tile;
    const int q = wino_locate(g, gt, tile);
    const float *x = g.src[0];
    const unsigned *xw = g.amax[0];
    WINO_SELECT(q, g, pr, x = g.src[i_]; xw = g.amax[i_];)
    const int H = pr.H, W = pr.W, TW = pr.TW;
    const int n = (int)(tile / (pr.TH * TW));
    const int r = (int)(tile - (int64_t)n * pr.TH * TW);
    const int th = r / TW, tw = r - th * TW;
    const int h0 = 4 * th - 1, w0 = 4 * tw - 1;
    const float *xb = x + (int64_t)n * H * W * C + c4;
    if (v_row_amax != nullptr) {
        const int e_img = rn_amax_exp_lanes(xw != nullptr ? reinterpret_cast<const unsigned char *>(xw) + (int64_t)n * RN_AMAX_BYTES : nullptr);
        if (gt_raw < t_last && c4 == 0) v_row_amax[t0 + gt] = rn_amax_word_of_exp(e_img, WINO_GAIN_BT);
    }
    if (gt_raw >= t_last) return;
    float4 t[6][6];                                           // t[i][j] = (B^T d)[i][j]
    float4 tz[6][4];                                          // tz[a][j] = (A dy)[a][j], dy = the patch's rows / columns 1..4
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float4 d[6], o[6];
        const int w = w0 + j;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int h = h0 + i;
            const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            d[i] = ok ? *reinterpret_cast<const float4 *>(xb + ((int64_t)h * W + w) * C) : f4(0.f);
        }
        bt6(d, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i][j] = o[i];
        if (j >= 1 && j <= 4) {
            float4 oz[6];
            a6(d + 1, oz);
#pragma unroll
            for (int a = 0; a < 6; ++a) tz[a][j - 1] = oz[a];
        }
    }
    float *zb = Z + (t0 + gt) * C + c4;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float4 o[6];
        a6(tz[a], o);
#pragma unroll
        for (int b = 0; b < 6; ++b) st_stream(zb + (int64_t)(a * 6 + b) * Tpad * C, o[b]);
    }
    float *vb = V + (t0 + gt) * C + c4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float4 o[6];
        bt6(t[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) st_stream(vb + (int64_t)(i * 6 + j) * Tpad * C, o[j]);
    }
}

// dw[co][(r*3 + s)*Cin + ci] += (G^T dU G)[r][s],  dU [36][Cout][Ku] (Ku = Cin rounded up to 32)
__global__ void wino_dw_kernel(const float *__restrict__ dU, float *__restrict__ dw, int Cout, int Cin, int Ku, int Kpad) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)Cout * Cin) return;
    const int co = (int)(id / Cin), ci = (int)(id - (int64_t)co * Cin);
    const float *u = dU + (int64_t)co * Ku + ci;
    const int64_t ps = (int64_t)Cout * Ku;
    float t[3][6];                                            // t[r][j] = sum_i G[i][r] dU[i][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = u[(i * 6 + j) * ps];
        t[0][j] = v[0] * (1.f / 4.f) - (v[1] + v[2]) * (1.f / 6.f) + (v[3] + v[4]) * (1.f / 24.f);
        t[1][j] = (v[2] - v[1]) * (1.f / 6.f) + (v[3] - v[4]) * (1.f / 12.f);
        t[2][j] = (v[3] + v[4] - v[1] - v[2]) * (1.f / 6.f) + v[5];
    }
    float *o = dw + (int64_t)co * Kpad + ci;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float *v = t[r];
        o[(r * 3 + 0) * Cin] += v[0] * (1.f / 4.f) - (v[1] + v[2]) * (1.f / 6.f) + (v[3] + v[4]) * (1.f / 24.f);
        o[(r * 3 + 1) * Cin] += (v[2] - v[1]) * (1.f / 6.f) + (v[3] - v[4]) * (1.f / 12.f);
        o[(r * 3 + 2) * Cin] += (v[3] + v[4] - v[1] - v[2]) * (1.f / 6.f) + v[5];
    }
}

// Host side: validate a group and build the device table.
static int wino_table(const rn_wino_group *g, WinoTable &t) {
    if (!g || g->n < 1 || g->n > RN_MAX_GROUP) return RN_EINVAL;
    t.n = g->n;
    int64_t end = 0;
    for (int i = 0; i < RN_MAX_GROUP; ++i) {
        const int k = i < g->n ? i : g->n - 1;               // unused slots repeat the last problem (never selected)
        if (g->N[k] <= 0 || g->H[k] <= 0 || g->W[k] <= 0) return RN_EINVAL;
        t.p[i].N = g->N[k]; t.p[i].H = g->H[k]; t.p[i].W = g->W[k];
        t.p[i].TH = (g->H[k] + 3) / 4; t.p[i].TW = (g->W[k] + 3) / 4;
        if (i < g->n) end += (int64_t)g->N[k] * t.p[i].TH * t.p[i].TW;
        t.tile_end[i] = end;
        t.src[i] = g->src[k]; t.dst[i] = g->dst[k]; t.add[i] = g->add[k]; t.mask[i] = g->mask[k];
        t.sign[i] = reinterpret_cast<unsigned *>(g->sign[k]);
        t.amax[i] = reinterpret_cast<unsigned *>(g->amax[k]);
    }
    return RN_OK;
}

extern "C" int rn_wino_input_group(const rn_wino_group *g, float *V, int C, int64_t tile_offset, int64_t Tpad, int dy_form,
                                   void *row_amax, void *tensor_amax, void *stream) {
    WinoTable t;
    const int rc = wino_table(g, t);
    if (rc) return rc;
    if (C <= 0 || (C & 3) || tile_offset < 0 || tile_offset + t.tile_end[t.n - 1] > Tpad) return RN_EINVAL;
    for (int i = 0; i < t.n; ++i)
        if (!t.src[i]) return RN_EINVAL;
    const dim3 grid(rn_blocks(t.tile_end[t.n - 1] * (C >> 2), 256));
    if (dy_form && row_amax != nullptr) return RN_EINVAL;    // A dy A^T feeds the weight gradient only: the tensor word
    if (dy_form) hipLaunchKernelGGL(wino_dy_kernel, grid, dim3(256), 0, (hipStream_t)stream, t, V, C, tile_offset, Tpad,
                                    reinterpret_cast<unsigned *>(tensor_amax));
    else hipLaunchKernelGGL(wino_in_kernel, grid, dim3(256), 0, (hipStream_t)stream, t, V, C, tile_offset, Tpad,
                            reinterpret_cast<unsigned *>(row_amax), reinterpret_cast<unsigned *>(tensor_amax));
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_wino_input_both_group(const rn_wino_group *g, float *V, float *Z, int C, int64_t tile_offset, int64_t Tpad,
                                        void *v_row_amax, void *z_tensor_amax, void *stream) {
    WinoTable t;
    const int rc = wino_table(g, t);
    if (rc) return rc;
    if (!V || !Z || C <= 0 || (C & 3) || tile_offset < 0 || tile_offset + t.tile_end[t.n - 1] > Tpad) return RN_EINVAL;
    for (int i = 0; i < t.n; ++i)
        if (!t.src[i]) return RN_EINVAL;
    hipLaunchKernelGGL(wino_in_both_kernel, dim3(rn_blocks(t.tile_end[t.n - 1] * (C >> 2), 256)), dim3(256), 0, (hipStream_t)stream, t,
                       V, Z, C, tile_offset, Tpad, reinterpret_cast<unsigned *>(v_row_amax), reinterpret_cast<unsigned *>(z_tensor_amax));
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_wino_output_group(const rn_wino_group *g, const float *M, int Cout, int64_t tile_offset, int64_t Tpad,
                                    const float *scale, const float *shift, int mask_mode, int act, int64_t y_batch_stride,
                                    void *stream) {
    WinoTable t;
    const int rc = wino_table(g, t);
    if (rc) return rc;
    if (Cout <= 0 || (Cout & 3) || tile_offset < 0 || tile_offset + t.tile_end[t.n - 1] > Tpad) return RN_EINVAL;
    if (mask_mode < 0 || (mask_mode & ~(3 | RN_MASK_BITS)) || (mask_mode & 3) == 3 || mask_mode == RN_MASK_BITS) return RN_EINVAL;
    if (act < 0 || act > 2 || y_batch_stride < 0 || (y_batch_stride & 3)) return RN_EINVAL;
    for (int i = 0; i < t.n; ++i) {
        if (!t.dst[i] || (mask_mode != 0) != (t.mask[i] != nullptr)) return RN_EINVAL;
        if (((mask_mode & RN_MASK_BITS) || t.sign[i]) && (Cout & 31)) return RN_EINVAL;     // whole words per pixel
        if (y_batch_stride && y_batch_stride < (int64_t)t.p[i].H * t.p[i].W * Cout) return RN_EINVAL;
    }
    // the instance that requests a launch's mask bits / addends in one batch (wino_out_kernel): addends only when every problem has one
    const bool pbits = (mask_mode & RN_MASK_BITS) != 0;
    bool padd = true;
    for (int i = 0; i < t.n; ++i) padd = padd && t.add[i] != nullptr;
    const dim3 grid(rn_blocks(t.tile_end[t.n - 1] * (Cout >> 2), 256));
#define RN_WINO_OUT(B, A) hipLaunchKernelGGL((wino_out_kernel<B, A>), grid, dim3(256), 0, (hipStream_t)stream, t, M, Cout, tile_offset, Tpad, \
                                             scale, shift, mask_mode, act, y_batch_stride)
    if (pbits && padd) RN_WINO_OUT(true, true);
    else if (pbits) RN_WINO_OUT(true, false);
    else if (padd) RN_WINO_OUT(false, true);
    else RN_WINO_OUT(false, false);
#undef RN_WINO_OUT
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// Single-problem forms.
static rn_wino_group wino_single(const float *src, float *dst, const float *add, const float *mask, int N, int H, int W) {
    rn_wino_group g = {};
    g.n = 1; g.N[0] = N; g.H[0] = H; g.W[0] = W; g.src[0] = src; g.dst[0] = dst; g.add[0] = add; g.mask[0] = mask;
    return g;
}
extern "C" int rn_wino_input(const float *x, float *V, int N, int H, int W, int C, int64_t tile_offset, int64_t Tpad,
                             void *stream) {
    const rn_wino_group g = wino_single(x, nullptr, nullptr, nullptr, N, H, W);
    return rn_wino_input_group(&g, V, C, tile_offset, Tpad, 0, nullptr, nullptr, stream);
}
extern "C" int rn_wino_dy(const float *dy, float *Z, int N, int H, int W, int C, int64_t tile_offset, int64_t Tpad, void *stream) {
    const rn_wino_group g = wino_single(dy, nullptr, nullptr, nullptr, N, H, W);
    return rn_wino_input_group(&g, Z, C, tile_offset, Tpad, 1, nullptr, nullptr, stream);
}
extern "C" int rn_wino_output(const float *M, float *y, int N, int H, int W, int Cout, int64_t tile_offset, int64_t Tpad,
                              const float *scale, const float *shift, const float *add, const float *mask, int mask_mode,
                              int act, int64_t y_batch_stride, void *stream) {
    const rn_wino_group g = wino_single(nullptr, y, add, mask, N, H, W);
    return rn_wino_output_group(&g, M, Cout, tile_offset, Tpad, scale, shift, mask_mode, act, y_batch_stride, stream);
}

extern "C" int rn_wino_weights(const float *w, float *U, int Cout, int Cin, int mode, const float *scale, void *stream) {
    if (Cout <= 0 || Cin <= 0 || mode < 0 || mode > 1) return RN_EINVAL;
    const int rows = mode == 0 ? Cout : Cin;
    const int Kpad = ((mode == 0 ? Cin : Cout) + 31) / 32 * 32;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(rn_blocks((int64_t)rows * Kpad, 256)), dim3(256), 0, (hipStream_t)stream, w, U,
                       Cout, Cin, mode, scale, rows, Kpad);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_wino_dw(const float *dU, float *dw, int Cout, int Cin, void *stream) {
    if (Cout <= 0 || Cin <= 0) return RN_EINVAL;
    const int Ku = (Cin + 31) / 32 * 32, Kpad = (9 * Cin + 31) / 32 * 32;
    hipLaunchKernelGGL(wino_dw_kernel, dim3(rn_blocks((int64_t)Cout * Cin, 256)), dim3(256), 0, (hipStream_t)stream, dU, dw, Cout,
                       Cin, Ku, Kpad);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
