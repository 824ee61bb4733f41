// Data-movement and per-channel kernels around the convolution engine (all NHWC fp32, HBM-bound).
//
// Replaces, in the reference: MaxPool2d(3,2,1) (D/model.py:216), the eval-mode BatchNorm2d arithmetic that is
// folded into conv epilogues here (D/model.py:278-282), nn.Upsample's backward in the FPN (D/model.py:88,99),
// Sigmoid's backward (D/model.py:180) and the autograd reductions that produce bias / batch-norm parameter
// gradients.  Weight re-packing (OIHW state_dict layout -> K-contiguous GEMM rows) has no reference counterpart:
// it is the derived cache SURVEY.md 8b asks to keep behind the unchanged parameter layout.
#include "common.h"
#include "mfma_split.h"

// ------------------------------------------------------------------------------------------------ weight packing
__global__ void pack_weights_kernel(const float *__restrict__ src, float *__restrict__ dst, int Cout, int Cin, int kh, int kw,
                                    int kw_pad, int c_pad, int mode, const float *__restrict__ scale, int rows, int Kpad,
                                    int r0, int nr, int s0, int ns) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * Kpad) return;
    const int row = (int)(i / Kpad), k = (int)(i - (int64_t)row * Kpad);
    const int tap = k / c_pad, c = k - tap * c_pad;
    int r = tap / kw_pad, s = tap - r * kw_pad;
    bool in_filter = r < kh && s < kw;
    if (mode == 2) {                        // tap subset: packed filter is nr x ns, source taps r0 + 2i, s0 + 2j
        r = tap / ns;
        s = tap - r * ns;
        in_filter = r < nr;
        r = r0 + 2 * r;
        s = s0 + 2 * s;
    }
    float v = 0.f;
    if (in_filter) {
        if (mode == 0) {                    // row = co, c = ci
            if (c < Cin) v = src[(((int64_t)row * Cin + c) * kh + r) * kw + s];
        } else {                            // dgrad layouts: row = ci, c = co
            if (c < Cout) {
                v = src[(((int64_t)c * Cin + row) * kh + r) * kw + s];
                if (scale) v *= scale[c];
            }
        }
    }
    dst[i] = v;
}

extern "C" int rn_pack_weights(const float *src, float *dst, int Cout, int Cin, int kh, int kw, int kw_pad, int c_pad,
                               int mode, const float *scale, int r0, int nr, int s0, int ns, void *stream) {
    if (Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || kw_pad < kw || (c_pad & 3) || mode < 0 || mode > 2) return RN_EINVAL;
    if (c_pad < (mode == 0 ? Cin : Cout)) return RN_EINVAL;
    if (mode == 2 && (nr <= 0 || ns <= 0 || r0 < 0 || s0 < 0 || r0 + 2 * (nr - 1) >= kh || s0 + 2 * (ns - 1) >= kw))
        return RN_EINVAL;
    const int rows = mode == 0 ? Cout : Cin;
    const int Kpad = ((mode == 2 ? nr * ns : kh * kw_pad) * c_pad + 31) / 32 * 32;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(rn_blocks((int64_t)rows * Kpad, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, Cout, Cin, kh, kw, kw_pad, c_pad, mode, scale, rows, Kpad, r0, nr, s0, ns);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// one workgroup per output channel: OIHW gradient + the folded batch-norm parameter gradients
__device__ __forceinline__ void unpack_wgrad_channel(const int co, const float *__restrict__ dw, const float *__restrict__ wp,
                                                     float *__restrict__ dweight, int Cin, int kh, int kw, int kw_pad,
                                                     int c_pad, int Kpad, const float *__restrict__ scale,
                                                     const float *__restrict__ mean, const float *__restrict__ rstd,
                                                     const float *__restrict__ colsum, float *__restrict__ dgamma,
                                                     float *__restrict__ dbeta) {
    __shared__ double red[4];
    const float sc = scale ? scale[co] : 1.f;
    const float *row = dw + (int64_t)co * Kpad;
    const float *wrow = wp + (int64_t)co * Kpad;
    const int n = Cin * kh * kw;
    double dot = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {            // i enumerates OIHW: ci, r, s
        const int ci = i / (kh * kw), rs = i - ci * (kh * kw);
        const int r = rs / kw, s = rs - r * kw;
        const int k = (r * kw_pad + s) * c_pad + ci;
        const float g = row[k];
        dweight[(int64_t)co * n + i] = sc * g;
        if (dgamma) dot += (double)wrow[k] * (double)g;
    }
    if (dgamma) {
        dot = wave_sum(dot);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = (red[0] + red[1]) + (red[2] + red[3]);
            dgamma[co] = (float)((t - (double)mean[co] * (double)colsum[co]) * (double)rstd[co]);
        }
    }
    if (dbeta && threadIdx.x == 0) dbeta[co] = colsum[co];
}

__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float *__restrict__ dw, const float *__restrict__ wp,
                                                           float *__restrict__ dweight, int Cin, int kh, int kw, int kw_pad,
                                                           int c_pad, int Kpad, const float *__restrict__ scale,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd,
                                                           const float *__restrict__ colsum, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta) {
    unpack_wgrad_channel(blockIdx.x, dw, wp, dweight, Cin, kh, kw, kw_pad, c_pad, Kpad, scale, mean, rstd, colsum, dgamma, dbeta);
}

// The same for many layers in one launch (the gradients of one all-reduce bucket, or of the whole net): ~70 launches of ~8 us per
// training step otherwise.  Block b does chunk b = (job, output channel) of a device-resident table.
__global__ __launch_bounds__(256) void unpack_batched_kernel(const rn_unpack_job *__restrict__ jobs, const int2 *__restrict__ chunks) {
    const int2 c = chunks[blockIdx.x];
    const rn_unpack_job j = jobs[c.x];
    unpack_wgrad_channel(c.y, j.dw, j.w_packed, j.dweight, j.Cin, j.kh, j.kw, j.kw_pad, j.c_pad, j.Kpad, j.scale, j.mean, j.rstd,
                         j.colsum, j.dgamma, j.dbeta);
}

extern "C" int rn_unpack_batched(const rn_unpack_job *jobs_dev, const int32_t *chunks_dev, int nchunks, void *stream) {
    if (!jobs_dev || !chunks_dev || nchunks <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(unpack_batched_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, jobs_dev,
                       reinterpret_cast<const int2 *>(chunks_dev));
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_unpack_wgrad(const float *dw, const float *w_packed, float *dweight, int Cout, int Cin, int kh, int kw,
                               int kw_pad, int c_pad, const float *scale, const float *mean, const float *rstd,
                               const float *colsum, float *dgamma, float *dbeta, void *stream) {
    if (Cout <= 0 || Cin <= 0 || kw_pad < kw || c_pad < Cin) return RN_EINVAL;
    if ((dgamma || dbeta) && !colsum) return RN_EINVAL;
    if (dgamma && (!mean || !rstd || !w_packed)) return RN_EINVAL;
    const int Kpad = (kh * kw_pad * c_pad + 31) / 32 * 32;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, dw, w_packed, dweight, Cin, kh, kw,
                       kw_pad, c_pad, Kpad, scale, mean, rstd, colsum, dgamma, dbeta);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ void bn_fold_kernel(const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mean,
                               const float *__restrict__ var, float eps, int C, float *__restrict__ scale,
                               float *__restrict__ shift, float *__restrict__ rstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float rs = 1.0f / sqrtf(var[c] + eps);
    const float s = gamma[c] * rs;
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
    if (rstd) rstd[c] = rs;
}

extern "C" int rn_bn_fold(const float *gamma, const float *beta, const float *mean, const float *var, float eps, int C,
                          float *scale, float *shift, float *rstd, void *stream) {
    if (C <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(bn_fold_kernel, dim3(rn_blocks(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps,
                       C, scale, shift, rstd);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ batched preparation
// A training step re-packs every weight tensor and re-folds every batch norm (the parameters have just changed):
// ~230 launches of a few microseconds each.  Here they are TWO launches over a device-resident job table (the
// optimizer's pattern, optim.hip): block b does chunk b = (job, block inside the job).  Two, because the data-gradient
// packs read the batch-norm scale the first launch produces.
__global__ __launch_bounds__(256) void prep_batched_kernel(const rn_prep_job *__restrict__ jobs, const int2 *__restrict__ chunks) {
    const int2 c = chunks[blockIdx.x];
    const rn_prep_job j = jobs[c.x];
    const int64_t i = (int64_t)c.y * 256 + threadIdx.x;
    if (j.kind == 0) {                                       // batch-norm folding, C = Cout
        if (i >= j.Cout) return;
        const float rs = 1.0f / sqrtf(j.var[i] + j.eps);
        const float sc = j.gamma[i] * rs;
        j.bn_scale[i] = sc;
        j.bn_shift[i] = j.beta[i] - j.mean[i] * sc;
        j.bn_rstd[i] = rs;
        return;
    }
    if (j.kind == 5) {                                       // fp16 pre-split form (RN_FP32_SPLIT3): a block = 4 rows, one wave each; bn_scale <- inverse row scales
        const int64_t row = (int64_t)c.y * 4 + (threadIdx.x >> 6);
        if (row < j.rows) split_row_f16(j.src, j.dst, j.bn_scale, row, j.Kpad, threadIdx.x & 63);
        return;
    }
    if (j.kind == 4) {                                       // pre-split form of an already packed fp32 buffer (mfma_split.h): 8 values per thread
        if (i < (int64_t)j.rows * j.Kpad / 8) split_store_chunk(j.src, j.dst, i);
        return;
    }
    if (i >= (int64_t)j.rows * j.Kpad) return;
    if (j.kind == 3) {                                       // bf16 copy of an already packed fp32 buffer (conv_bf16.hip): src -> dst
        reinterpret_cast<__bf16 *>(j.dst)[i] = (__bf16)j.src[i];
        return;
    }
    if (j.kind == 2) {                                       // Winograd weight transform U = G g G^T (conv_wino.hip), mode 0 / 1
        const int row = (int)(i / j.Kpad), k = (int)(i - (int64_t)row * j.Kpad);
        const int kdim = j.mode == 0 ? j.Cin : j.Cout;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float v = 0.f;
                if (k < kdim) {
                    if (j.mode == 0) v = j.src[(((int64_t)row * j.Cin + k) * 3 + r) * 3 + s];
                    else v = j.src[(((int64_t)k * j.Cin + row) * 3 + (2 - r)) * 3 + (2 - s)] * (j.scale ? j.scale[k] : 1.f);
                }
                g[r][s] = v;
            }
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
        for (int a6i = 0; a6i < 6; ++a6i) {
            const float a = t[a6i][0], b = t[a6i][1], c = t[a6i][2];
            const float o[6] = {a * (1.f / 4.f), -(a + b + c) * (1.f / 6.f), (b - a - c) * (1.f / 6.f),
                                a * (1.f / 24.f) + b * (1.f / 12.f) + c * (1.f / 6.f),
                                a * (1.f / 24.f) - b * (1.f / 12.f) + c * (1.f / 6.f), c};
#pragma unroll
            for (int b6 = 0; b6 < 6; ++b6) j.dst[((int64_t)(a6i * 6 + b6) * j.rows + row) * j.Kpad + k] = o[b6];
        }
        return;
    }
    // weight packing: the body of pack_weights_kernel
    const int row = (int)(i / j.Kpad), k = (int)(i - (int64_t)row * j.Kpad);
    const int tap = k / j.c_pad, ch = k - tap * j.c_pad;
    int r = tap / j.kw_pad, s = tap - r * j.kw_pad;
    bool in_filter = r < j.kh && s < j.kw;
    if (j.mode == 2) {
        r = tap / j.ns;
        s = tap - r * j.ns;
        in_filter = r < j.nr;
        r = j.r0 + 2 * r;
        s = j.s0 + 2 * s;
    }
    float v = 0.f;
    if (in_filter) {
        if (j.mode == 0) {
            if (ch < j.Cin) v = j.src[(((int64_t)row * j.Cin + ch) * j.kh + r) * j.kw + s];
        } else if (ch < j.Cout) {
            v = j.src[(((int64_t)ch * j.Cin + row) * j.kh + r) * j.kw + s];
            if (j.scale) v *= j.scale[ch];
        }
    }
    j.dst[i] = v;
}

extern "C" int rn_prep_batched(const rn_prep_job *jobs_dev, const int32_t *chunks_dev, int nchunks, void *stream) {
    if (!jobs_dev || !chunks_dev || nchunks <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(prep_batched_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, jobs_dev,
                       reinterpret_cast<const int2 *>(chunks_dev));
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ layout
// (every kernel below that produces an activation / gradient tensor takes `amax`: NULL, or the result's amax words, one per image --
// include/retinanet_mi355x.h: rn_conv_desc.y_amax; mfma_split.h: rn_amax_note)
__global__ void nchw_to_nhwc4_kernel(const float *__restrict__ src, float4 *__restrict__ dst, int64_t HW, int64_t total, unsigned *__restrict__ amax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // pixel index over N*H*W
    if (i >= total) return;
    const int64_t n = i / HW, p = i - n * HW;
    const float *s = src + n * 3 * HW + p;
    const float4 v = make_float4(s[0], s[HW], s[2 * HW], 0.f);
    dst[i] = v;
    rn_amax_note(amax, n, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)));
}

extern "C" int rn_nchw_to_nhwc4(const float *src, float *dst, int N, int H, int W, void *amax, void *stream) {
    if (N <= 0 || H <= 0 || W <= 0) return RN_EINVAL;
    const int64_t HW = (int64_t)H * W, total = HW * N;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<float4 *>(dst), HW, total, reinterpret_cast<unsigned *>(amax));
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ max-pool 3x3 s2 p1
__global__ void maxpool_fwd_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, uchar4 *__restrict__ arg, int H, int W,
                                   int C4, int Ho, int Wo, int64_t total) {
    // (3x3 / stride 2 windows overlap: one contiguous band of output rows per XCD, so the shared input rows meet in one L2)
    const int64_t i = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;       // over N*Ho*Wo*C4
    if (i >= total) return;
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int64_t n = t / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 a = make_uchar4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ih = oh * 2 - 1 + r;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int iw = ow * 2 - 1 + s;
            if ((unsigned)iw >= (unsigned)W) continue;
            const float4 v = x[((n * H + ih) * W + iw) * C4 + c];
            const unsigned char pos = (unsigned char)(3 * r + s);     // strict > keeps the first maximum
            if (v.x > m.x) { m.x = v.x; a.x = pos; }
            if (v.y > m.y) { m.y = v.y; a.y = pos; }
            if (v.z > m.z) { m.z = v.z; a.z = pos; }
            if (v.w > m.w) { m.w = v.w; a.w = pos; }
        }
    }
    y[i] = m;
    if (arg) arg[i] = a;
}

extern "C" int rn_maxpool_fwd(const float *x, float *y, uint8_t *argmax, int N, int H, int W, int C, int Ho, int Wo,
                              void *stream) {
    if (N <= 0 || (C & 3) || Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return RN_EINVAL;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y), reinterpret_cast<uchar4 *>(argmax), H,
                       W, C / 4, Ho, Wo, total);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// Gather form: every input element looks at the (at most 4) windows that contain it and takes dy of those whose
// recorded first maximum is this element.
__global__ void maxpool_bwd_kernel(const float4 *__restrict__ x, const float4 *__restrict__ dy, const uchar4 *__restrict__ arg,
                                   float4 *__restrict__ dx, int H, int W, int C4, int Ho, int Wo, int relu_mask, int64_t total,
                                   unsigned *__restrict__ amax) {
    const int64_t i = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;       // over N*H*W*C4 (band per XCD, as above)
    if (i >= total) return;
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int iw = (int)(t % W);
    t /= W;
    const int ih = (int)(t % H);
    const int64_t n = t / H;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    // windows oh with 2*oh-1 <= ih <= 2*oh+1, i.e. ih/2 <= oh <= (ih+1)/2
    for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
        if (oh >= Ho) continue;
        const int r = ih - (oh * 2 - 1);
        for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
            if (ow >= Wo) continue;
            const unsigned char pos = (unsigned char)(3 * r + (iw - (ow * 2 - 1)));
            const int64_t o = ((n * Ho + oh) * Wo + ow) * C4 + c;
            const uchar4 a = arg[o];
            const float4 d = dy[o];
            if (a.x == pos) g.x += d.x;
            if (a.y == pos) g.y += d.y;
            if (a.z == pos) g.z += d.z;
            if (a.w == pos) g.w += d.w;
        }
    }
    if (relu_mask) {                                            // 1: x is the fp32 activation; 2: x points to its sign bits (common.h)
        const float4 v = rn_mask_load4(reinterpret_cast<const float *>(x), 4 * i, relu_mask == 2);
        g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
    }
    dx[i] = g;
    rn_amax_note(amax, n, fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))));
}

extern "C" int rn_maxpool_bwd(const float *x, const float *dy, const uint8_t *argmax, float *dx, int N, int H, int W, int C,
                              int Ho, int Wo, int relu_mask, void *amax, void *stream) {
    if (N <= 0 || (C & 3) || argmax == nullptr || Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return RN_EINVAL;
    if (relu_mask < 0 || relu_mask > 2 || (relu_mask == 2 && (C & 31))) return RN_EINVAL;
    const int64_t total = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(dy),
                       reinterpret_cast<const uchar4 *>(argmax), reinterpret_cast<float4 *>(dx), H, W, C / 4, Ho, Wo, relu_mask,
                       total, reinterpret_cast<unsigned *>(amax));
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ column sums
#define CS_ROWS 512            // rows per first-pass workgroup
extern "C" int64_t rn_colsum_workspace_bytes(int64_t rows, int C) {
    return ((rows + CS_ROWS - 1) / CS_ROWS) * (int64_t)C * sizeof(float);
}

// pass 1: workgroup (chunk of rows) x (256 columns): lane = column, loop over rows -> coalesced row reads
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ g, int64_t rows, int C, int ld,
                                                             float *__restrict__ part) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS;
    const int64_t r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += g[r * ld + c]; s1 += g[(r + 1) * ld + c]; s2 += g[(r + 2) * ld + c]; s3 += g[(r + 3) * ld + c];
    }
    for (; r < r1; ++r) s0 += g[r * ld + c];
    part[(int64_t)blockIdx.x * C + c] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ part, int64_t nchunks, int C,
                                                           float *__restrict__ out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = accumulate ? (double)out[c] : 0.0;
    for (int64_t k = 0; k < nchunks; ++k) s += part[k * C + c];
    out[c] = (float)s;
}

extern "C" int rn_colsum(const float *g, int64_t rows, int C, int ld, float *out, int accumulate, void *workspace,
                         void *stream) {
    if (rows <= 0 || C <= 0 || ld < C) return RN_EINVAL;
    const int64_t nchunks = (rows + CS_ROWS - 1) / CS_ROWS;
    float *part = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nchunks, (C + 255) / 256), dim3(256), 0, (hipStream_t)stream, g,
                       rows, C, ld, part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float *)part,
                       nchunks, C, out, accumulate);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ FPN top-down backward
__global__ void upsample_add_bwd_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, int Hs, int Ws, int Hd, int Wd,
                                        int C4, int64_t total, unsigned *__restrict__ amax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // over N*Hd*Wd*C4
    if (i >= total) return;
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int w = (int)(t % Wd);
    t /= Wd;
    const int h = (int)(t % Hd);
    const int64_t n = t / Hd;
    float4 a = dst[i];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int hs = 2 * h + dy;
        if (hs >= Hs) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ws = 2 * w + dx;
            if (ws >= Ws) continue;
            const float4 v = src[((n * Hs + hs) * Ws + ws) * C4 + c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    dst[i] = a;
    rn_amax_note(amax, n, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
}

extern "C" int rn_upsample_add_bwd(const float *src, float *dst, int N, int Hs, int Ws, int Hd, int Wd, int C, void *amax, void *stream) {
    if (N <= 0 || (C & 3)) return RN_EINVAL;
    const int64_t total = (int64_t)N * Hd * Wd * (C / 4);
    hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), Hs, Ws, Hd, Wd, C / 4, total,
                       reinterpret_cast<unsigned *>(amax));
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ small elementwise
__global__ void relu_mask_kernel(float *__restrict__ g, const float *__restrict__ z, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = z[i] > 0.f ? g[i] : 0.f;
}
extern "C" int rn_relu_mask(float *g, const float *z, int64_t n, void *stream) {
    if (n <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(rn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, g, z, n);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ void sigmoid_bwd_pad_kernel(const float *__restrict__ dy, const float *__restrict__ s, float *__restrict__ out,
                                       int64_t rows, int64_t rpi, int C, int ld, int64_t bstride, unsigned *__restrict__ amax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ld) return;
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    const int64_t b = r / rpi;
    float v = 0.f;
    if (c < C) {
        const int64_t src = b * bstride + (r - b * rpi) * C + c;
        v = dy[src];
        if (s) { const float p = s[src]; v *= p * (1.0f - p); }
    }
    out[i] = v;
    rn_amax_note(amax, b, fabsf(v));
}
extern "C" int rn_sigmoid_bwd_pad(const float *dy, const float *s, float *out, int B, int64_t rows_per_image, int C, int ld,
                                  int64_t src_batch_stride, void *amax, void *stream) {
    const int64_t rows = (int64_t)B * rows_per_image;
    if (rows <= 0 || C <= 0 || ld < C) return RN_EINVAL;
    hipLaunchKernelGGL(sigmoid_bwd_pad_kernel, dim3(rn_blocks(rows * ld, 256)), dim3(256), 0, (hipStream_t)stream, dy, s, out,
                       rows, rows_per_image, C, ld, src_batch_stride, reinterpret_cast<unsigned *>(amax));
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ void add_inplace_kernel(float *__restrict__ dst, const float *__restrict__ src, int64_t n, int64_t per_image, unsigned *__restrict__ amax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t img = amax != nullptr ? i / per_image : 0;
    const float v = dst[i] + src[i];
    dst[i] = v;
    rn_amax_note(amax, img, fabsf(v));
}
extern "C" int rn_add_inplace(float *dst, const float *src, int64_t n, int64_t per_image, void *amax, void *stream) {
    if (n <= 0 || (amax != nullptr && per_image <= 0)) return RN_EINVAL;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(rn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n, per_image,
                       reinterpret_cast<unsigned *>(amax));
    RN_LAUNCH_CHECK();
    return RN_OK;
}
