// The non-convolution steps of the training schedule for bf16 activations (conv_bf16.hip; BASELINE configs[2]).
// Same operations as elementwise.hip (MaxPool2d 3x3/2 of D/model.py:216 and its gradient, the gradient of the FPN's
// nearest-upsample + add, D/model.py:88-108, the head-output gradient slices with the sigmoid derivative, D/model.py:196),
// with bf16 storage on the activation side; arithmetic in fp32, one rounding on the store.  The stem stays fp32 (3-channel
// input), so the pooling kernels are the fp32 <-> bf16 boundary: pool forward reads the fp32 stem output and writes bf16,
// pool backward reads a bf16 gradient and writes the fp32 gradient of the stem output.
// Roofline: HBM (streaming).
#include <math.h>

#include "common.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_bf4(const __bf16 *p) {
    const bf16x4 q = *reinterpret_cast<const bf16x4 *>(p);
    return make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
}
__device__ __forceinline__ void st_bf4(__bf16 *p, float4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4 *>(p) = o;
}

__global__ void maxpool_fwd_bf16out_kernel(const float4 *__restrict__ x, __bf16 *__restrict__ y, uchar4 *__restrict__ arg, int H,
                                           int W, int C4, int Ho, int Wo, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // over N*Ho*Wo*C4
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
    st_bf4(y + i * 4, m);
    if (arg) arg[i] = a;
}
extern "C" int rn_maxpool_fwd_bf16out(const float *x, void *y, uint8_t *argmax, int N, int H, int W, int C, int Ho, int Wo,
                                      void *stream) {
    if (N <= 0 || (C & 3) || Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return RN_EINVAL;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_bf16out_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<__bf16 *>(y), reinterpret_cast<uchar4 *>(argmax), H, W,
                       C / 4, Ho, Wo, total);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ void maxpool_bwd_bf16in_kernel(const float4 *__restrict__ x, const __bf16 *__restrict__ dy, const uchar4 *__restrict__ arg,
                                          float4 *__restrict__ dx, int H, int W, int C4, int Ho, int Wo, int relu_mask, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // over N*H*W*C4
    if (i >= total) return;
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int iw = (int)(t % W);
    t /= W;
    const int ih = (int)(t % H);
    const int64_t n = t / H;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
        if (oh >= Ho) continue;
        const int r = ih - (oh * 2 - 1);
        for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
            if (ow >= Wo) continue;
            const unsigned char pos = (unsigned char)(3 * r + (iw - (ow * 2 - 1)));
            const int64_t o = ((n * Ho + oh) * Wo + ow) * C4 + c;
            const uchar4 a = arg[o];
            const float4 d = ld_bf4(dy + o * 4);
            if (a.x == pos) g.x += d.x;
            if (a.y == pos) g.y += d.y;
            if (a.z == pos) g.z += d.z;
            if (a.w == pos) g.w += d.w;
        }
    }
    if (relu_mask) {                                            // 1: x is the fp32 stem output; 2: x points to its sign bits (common.h)
        const float4 v = rn_mask_load4(reinterpret_cast<const float *>(x), 4 * i, relu_mask == 2);
        g.x = v.x > 0.f ? g.x : 0.f; g.y = v.y > 0.f ? g.y : 0.f; g.z = v.z > 0.f ? g.z : 0.f; g.w = v.w > 0.f ? g.w : 0.f;
    }
    dx[i] = g;
}
extern "C" int rn_maxpool_bwd_bf16in(const float *x, const void *dy, const uint8_t *argmax, float *dx, int N, int H, int W, int C,
                                     int Ho, int Wo, int relu_mask, void *stream) {
    if (N <= 0 || (C & 3) || argmax == nullptr || Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return RN_EINVAL;
    if (relu_mask < 0 || relu_mask > 2 || (relu_mask == 2 && (C & 31))) return RN_EINVAL;
    const int64_t total = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_bf16in_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<const __bf16 *>(dy),
                       reinterpret_cast<const uchar4 *>(argmax), reinterpret_cast<float4 *>(dx), H, W, C / 4, Ho, Wo, relu_mask, total);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ void upsample_add_bwd_bf16_kernel(const __bf16 *__restrict__ src, __bf16 *__restrict__ dst, int Hs, int Ws, int Hd, int Wd,
                                             int C4, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // over N*Hd*Wd*C4
    if (i >= total) return;
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int w = (int)(t % Wd);
    t /= Wd;
    const int h = (int)(t % Hd);
    const int64_t n = t / Hd;
    float4 a = ld_bf4(dst + i * 4);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int hs = 2 * h + dy;
        if (hs >= Hs) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ws = 2 * w + dx;
            if (ws >= Ws) continue;
            const float4 v = ld_bf4(src + (((n * Hs + hs) * Ws + ws) * C4 + c) * 4);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    st_bf4(dst + i * 4, a);
}
extern "C" int rn_upsample_add_bwd_bf16(const void *src, void *dst, int N, int Hs, int Ws, int Hd, int Wd, int C, void *stream) {
    if (N <= 0 || (C & 3)) return RN_EINVAL;
    const int64_t total = (int64_t)N * Hd * Wd * (C / 4);
    hipLaunchKernelGGL(upsample_add_bwd_bf16_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16 *>(src), reinterpret_cast<__bf16 *>(dst), Hs, Ws, Hd, Wd, C / 4, total);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ void sigmoid_bwd_pad_bf16_kernel(const float *__restrict__ dy, const float *__restrict__ s, __bf16 *__restrict__ out,
                                            int64_t rows, int64_t rpi, int C, int ld, int64_t bstride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * ld) return;
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    float v = 0.f;
    if (c < C) {
        const int64_t b = r / rpi;
        const int64_t src = b * bstride + (r - b * rpi) * C + c;
        v = dy[src];
        if (s) { const float p = s[src]; v *= p * (1.0f - p); }
    }
    out[i] = (__bf16)v;
}
extern "C" int rn_sigmoid_bwd_pad_bf16(const float *dy, const float *s, void *out, int B, int64_t rows_per_image, int C, int ld,
                                       int64_t src_batch_stride, void *stream) {
    const int64_t rows = (int64_t)B * rows_per_image;
    if (rows <= 0 || C <= 0 || ld < C) return RN_EINVAL;
    hipLaunchKernelGGL(sigmoid_bwd_pad_bf16_kernel, dim3(rn_blocks(rows * ld, 256)), dim3(256), 0, (hipStream_t)stream, dy, s,
                       reinterpret_cast<__bf16 *>(out), rows, rows_per_image, C, ld, src_batch_stride);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ void relu_bf16_kernel(const __bf16 *__restrict__ src, __bf16 *__restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = ld_bf4(src + i * 4);
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    st_bf4(dst + i * 4, v);
}
extern "C" int rn_relu_bf16(const void *src, void *dst, int64_t n, void *stream) {
    if (n <= 0 || (n & 3)) return RN_EINVAL;
    hipLaunchKernelGGL(relu_bf16_kernel, dim3(rn_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16 *>(src), reinterpret_cast<__bf16 *>(dst), n / 4);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
