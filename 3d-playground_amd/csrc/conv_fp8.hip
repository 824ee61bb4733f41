// fp8 (OCP e4m3fn) implicit-GEMM convolution, forward / inference form -- BASELINE configs[4] ("fp8 weights / activations on the
// CDNA4 fp8 MFMA"), first cut: the fprop of every convolution of the detector behind the fp32 stem (D/model.py:59-205,
// D/utils.py:12-80; network of configs[4]: D/model.py:434-443) with its fused frozen batch-norm / bias / residual / ReLU /
// sigmoid / FPN upsample-add epilogue.  No backward pass, no weight gradient: training stays fp32 / bf16.
//
// Arithmetic: v_mfma_scale_f32_32x32x64_f8f6f4 with both operands e4m3 and every block scale 2^0 -- the scaled form is the one
// that runs at the fp8 rate (twice the bf16 MFMA per clock; the non-scaled fp8 MFMA runs at the bf16 rate:
// MI355X_MICROARCH.md, Matrix cores); fp32 accumulation.  Quantisation is by real-valued scales applied in the epilogue, where
// they cost nothing: weights per OUTPUT CHANNEL (w_q = fp8(w / sw[c]), sw[c] = max|w[c]| / 448, rn_fp8_quantize_rows),
// activations per TENSOR (x_q = fp8(x / sx), sx calibrated by the host logic against the bf16 / fp32 path):
//     y = act( acc * (sx * sw[c] * bn_scale[c]) + shift[c] + add_q * s_add ),   stored as fp8(y / sy) or as fp32.
// The caller folds sx * sw[c] * bn_scale[c] into `scale`.
//
// Tile / staging: the bf16 kernel's (conv_bf16.hip) BYTE for byte -- a staged row is 64 bytes of K (64 fp8 instead of 32 bf16),
// direct-to-LDS buffer loads, out-of-range offsets as zero padding, the XOR swizzle on the load's source and on the fragment read.
// One 32x32x64 operand = this lane's 32 consecutive K values of row lane & 31 (K half lane >> 5): two ds_read_b128.  Per 64-byte
// K-step a wave issues 8 reads against 4 MFMAs of 64 cycles: twice the FLOPs of the bf16 kernel on the same staged bytes.
//
// Roofline: MFMA by FLOPs (~5 PF dense fp8); in practice staging bandwidth, like the bf16 kernel.
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define F8_BK 64                 // K elements per step = 64 bytes per staged row
#define F8_MAX 448.0f            // largest finite e4m3fn

__device__ __forceinline__ int f8_swz(int row) { return (row >> 2) & 3; }

// two floats -> two e4m3 in the low / high half of a dword (round to nearest even, saturating by the clamp in front)
__device__ __forceinline__ float f8_clamp(float a) { return fminf(fmaxf(a, -F8_MAX), F8_MAX); }
__device__ __forceinline__ int f8_pack4(float a, float b, float c, float d) {
    const int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(a), f8_clamp(b), 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(c), f8_clamp(d), lo, true);
}

// ---------------------------------------------------------------------------------------------- quantisation
// dst[i] = fp8(src[i] * inv_scale): activations entering the fp8 part of the network (per-tensor scale).
__global__ void fp8_quantize_kernel(const float *__restrict__ src, int *__restrict__ dst, int64_t n4, float inv_scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4 *>(src)[i];
    dst[i] = f8_pack4(v.x * inv_scale, v.y * inv_scale, v.z * inv_scale, v.w * inv_scale);
}
extern "C" int rn_fp8_quantize(const float *src, void *dst, int64_t n, float inv_scale, void *stream) {
    if (n <= 0 || (n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 3)) return RN_EINVAL;
    hipLaunchKernelGGL(fp8_quantize_kernel, dim3(rn_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<int *>(dst), n / 4, inv_scale);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
__global__ void fp8_dequantize_kernel(const int *__restrict__ src, float *__restrict__ dst, int64_t n4, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int q = src[i];
    reinterpret_cast<float4 *>(dst)[i] = make_float4(__builtin_amdgcn_cvt_f32_fp8(q, 0) * scale, __builtin_amdgcn_cvt_f32_fp8(q, 1) * scale,
                                                     __builtin_amdgcn_cvt_f32_fp8(q, 2) * scale, __builtin_amdgcn_cvt_f32_fp8(q, 3) * scale);
}
extern "C" int rn_fp8_dequantize(const void *src, float *dst, int64_t n, float scale, void *stream) {
    if (n <= 0 || (n & 3) || ((uintptr_t)src & 3) || ((uintptr_t)dst & 15)) return RN_EINVAL;
    hipLaunchKernelGGL(fp8_dequantize_kernel, dim3(rn_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const int *>(src), dst, n / 4, scale);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// e4m3 -> bf16 with the tensor's scale (round 5: the fp8-forward TRAINING step keeps its activations as e4m3 and hands them to the bf16
// data / weight gradient kernels; 16 values per thread: one 16-byte load, two 16-byte stores).  q * scale is rounded once, to bf16.
typedef __bf16 rn_bf16x8 __attribute__((ext_vector_type(8)));
__global__ void fp8_to_bf16_kernel(const int4 *__restrict__ src, rn_bf16x8 *__restrict__ dst, int64_t n16, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const int4 q = src[i];
    const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        rn_bf16x8 o;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = w[2 * h + j];
            o[4 * j + 0] = (__bf16)(__builtin_amdgcn_cvt_f32_fp8(d, 0) * scale);
            o[4 * j + 1] = (__bf16)(__builtin_amdgcn_cvt_f32_fp8(d, 1) * scale);
            o[4 * j + 2] = (__bf16)(__builtin_amdgcn_cvt_f32_fp8(d, 2) * scale);
            o[4 * j + 3] = (__bf16)(__builtin_amdgcn_cvt_f32_fp8(d, 3) * scale);
        }
        dst[2 * i + h] = o;
    }
}
extern "C" int rn_fp8_to_bf16(const void *src, void *dst, int64_t n, float scale, void *stream) {
    if (n <= 0 || (n & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return RN_EINVAL;
    hipLaunchKernelGGL(fp8_to_bf16_kernel, dim3(rn_blocks(n / 16, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const int4 *>(src), reinterpret_cast<rn_bf16x8 *>(dst), n / 16, scale);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// bf16 -> e4m3: dst[i] = fp8(src[i] * inv_scale), saturating; 16 values per thread.  The fp8 engine's bf16 residual stream (round 5: the
// last convolution of a bottleneck and the FPN run in bf16, tools/fp8_error_budget.py) enters the next block's e4m3 convolutions here.
__global__ void bf16_to_fp8_kernel(const rn_bf16x8 *__restrict__ src, int4 *__restrict__ dst, int64_t n16, float inv_scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const rn_bf16x8 a = src[2 * i], b = src[2 * i + 1];
    int4 q;
    q.x = f8_pack4((float)a[0] * inv_scale, (float)a[1] * inv_scale, (float)a[2] * inv_scale, (float)a[3] * inv_scale);
    q.y = f8_pack4((float)a[4] * inv_scale, (float)a[5] * inv_scale, (float)a[6] * inv_scale, (float)a[7] * inv_scale);
    q.z = f8_pack4((float)b[0] * inv_scale, (float)b[1] * inv_scale, (float)b[2] * inv_scale, (float)b[3] * inv_scale);
    q.w = f8_pack4((float)b[4] * inv_scale, (float)b[5] * inv_scale, (float)b[6] * inv_scale, (float)b[7] * inv_scale);
    dst[i] = q;
}
extern "C" int rn_bf16_to_fp8(const void *src, void *dst, int64_t n, float inv_scale, void *stream) {
    if (n <= 0 || (n & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return RN_EINVAL;
    hipLaunchKernelGGL(bf16_to_fp8_kernel, dim3(rn_blocks(n / 16, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const rn_bf16x8 *>(src), reinterpret_cast<int4 *>(dst), n / 16, inv_scale);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// 3x3 / stride 2 / pad 1 max-pool of the fp32 stem output, written as e4m3 with one scale: what max-pool followed by rn_fp8_quantize
// computes (bit for bit), without the fp32 pooled tensor in between (D/model.py:232: the boundary where the fp8 engine's fp32 stem ends).
__global__ void maxpool_fwd_fp8out_kernel(const float4 *__restrict__ x, int *__restrict__ y, int H, int W, int C4, int Ho, int Wo,
                                          float inv_scale, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // over N*Ho*Wo*C4
    if (i >= total) return;
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int64_t n = t / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ih = oh * 2 - 1 + r;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            const int iw = ow * 2 - 1 + s_;
            if ((unsigned)iw >= (unsigned)W) continue;
            const float4 v = x[((n * H + ih) * W + iw) * C4 + c];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    y[i] = f8_pack4(m.x * inv_scale, m.y * inv_scale, m.z * inv_scale, m.w * inv_scale);
}
extern "C" int rn_maxpool_fwd_fp8out(const float *x, void *y, int N, int H, int W, int C, int Ho, int Wo, float inv_scale, void *stream) {
    if (N <= 0 || (C & 3) || Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1 || ((uintptr_t)x & 15) || ((uintptr_t)y & 3)) return RN_EINVAL;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_fp8out_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), reinterpret_cast<int *>(y), H, W, C / 4, Ho, Wo, inv_scale, total);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// Packed fp32 weight rows [rows][Kpad] (rn_pack_weights) -> fp8 rows [rows][Kpad64] (Kpad rounded up to 64: a K-step reads 64
// bytes of a row) with one scale per row: row_scale[r] = max|row| / 448 (1 for an all-zero row), dst = fp8(src / row_scale[r]).
// One workgroup per row.
__global__ __launch_bounds__(256) void fp8_quantize_rows_kernel(const float *__restrict__ src, unsigned char *__restrict__ dst,
                                                                float *__restrict__ row_scale, int Kpad, int Kpad64) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float *s = src + (int64_t)r * Kpad;
    float m = 0.f;
    for (int k = threadIdx.x; k < Kpad; k += 256) m = fmaxf(m, fabsf(s[k]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = m > 0.f ? m / F8_MAX : 1.f;
    if (threadIdx.x == 0) row_scale[r] = sc;
    const float inv = 1.f / sc;
    int *d = reinterpret_cast<int *>(dst + (int64_t)r * Kpad64);
    for (int k4 = threadIdx.x; k4 < Kpad64 / 4; k4 += 256) {
        const int k = 4 * k4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (k + j < Kpad) ? s[k + j] * inv : 0.f;
        d[k4] = f8_pack4(v[0], v[1], v[2], v[3]);
    }
}
extern "C" int rn_fp8_quantize_rows(const float *w_packed, void *w_q, float *row_scale, int64_t rows, int Kpad, void *stream) {
    if (rows <= 0 || Kpad <= 0 || (Kpad & 3) || ((uintptr_t)w_q & 15)) return RN_EINVAL;
    hipLaunchKernelGGL(fp8_quantize_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, w_packed,
                       reinterpret_cast<unsigned char *>(w_q), row_scale, Kpad, (Kpad + 63) / 64 * 64);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ---------------------------------------------------------------------------------------------- implicit GEMM
struct Fp8Args {
    float add_scale;             // the addend's per-tensor scale (its values are fp8)
    float out_inv_scale;         // 1 / the result's per-tensor scale (fp8 result), unused for an fp32 result
};

// 128 x 128 output tile by 2 x 2 waves, each a 64 x 64 sub-tile = 2 x 2 accumulators of the 32x32x64 MFMA.
template <bool YF32>
__device__ __forceinline__ void conv_igemm_fp8_tile(const rn_conv_desc &d, const unsigned char *__restrict__ x,
                                                    const unsigned char *__restrict__ w, void *__restrict__ yv,
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    const unsigned char *__restrict__ add, const Fp8Args fa_, const int tile) {
    constexpr int BM = 128, BN = 128, NW = 4;
    constexpr int RF = 16;                                   // 4-byte words per staged row: 64 bytes
    constexpr int RPI = 16;                                  // rows one wave instruction fills (1 KiB / 64 B)
    constexpr int IA = BM / RPI / NW, IB = BN / RPI / NW;    // DMA instructions per wave per K-step and operand
    constexpr int STEP = (BM + BN) * RF;                     // words per buffer: A rows, then B rows
    constexpr int LDT = BN + 4;
    constexpr int RP = 64;
    constexpr int NBUF = 3;
    constexpr int LDSF = NBUF * STEP > RP * LDT ? NBUF * STEP : RP * LDT;
    __shared__ float lds[LDSF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ntn = (d.Cout + BN - 1) / BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int HoWo = d.Ho * d.Wo;
    const int64_t M = (int64_t)d.N * HoWo;
    const int K = d.kh * d.kw * d.Cin;
    const int Kpad = (K + 63) / 64 * 64;                     // fp8 weight rows: zero-padded to a multiple of 64 (rn_fp8_quantize_rows)
    const int nks = Kpad / F8_BK;
    const int dmask = (1 << d.div_shift) - 1;

    const int n_first = (int)(m0 / HoWo);
    const int64_t x_elems = ((int64_t)d.N - 1 - n_first) * d.x_batch_stride + (int64_t)d.Hi * d.Wi * d.Cin;
    const v4i32 rs_a = make_rsrc(x + (int64_t)n_first * d.x_batch_stride, (unsigned)(x_elems > 0x7FFFFFFF ? 0x7FFFFFFF : x_elems));
    const v4i32 rs_b = make_rsrc(w, (unsigned)((int64_t)d.Cout * Kpad));

    // staging geometry: instruction j of this wave fills rows (wave*I + j)*16 .. +15; the lane fills 16-byte position pos of row
    // rsub, i.e. fetches logical chunk pos ^ swz(row) = 16 elements of K
    const int pos = lane & 3, rsub = lane >> 2;
    int a_h[IA], a_w[IA], a_img[IA], a_c[IA];
    unsigned a_voff[IA];
    const int rel0 = m0 - n_first * HoWo;
#pragma unroll
    for (int j = 0; j < IA; ++j) {
        const int row = (wave * IA + j) * RPI + rsub;
        a_c[j] = 16 * (pos ^ f8_swz(row));
        if ((int64_t)m0 + row < M) {
            const unsigned rel = (unsigned)(rel0 + row);
            const unsigned n = rel / (unsigned)HoWo;
            const unsigned rem = rel - n * (unsigned)HoWo;
            const unsigned oh = rem / (unsigned)d.Wo, ow = rem - oh * (unsigned)d.Wo;
            a_img[j] = (int)((int64_t)n * d.x_batch_stride);
            a_h[j] = (int)oh * d.a + d.p;
            a_w[j] = (int)ow * d.a + d.p_w;
        } else {
            a_img[j] = 0;
            a_h[j] = -(1 << 28);
            a_w[j] = 0;
        }
        a_voff[j] = 0x80000000u;
    }
    unsigned b_voff[IB];
#pragma unroll
    for (int j = 0; j < IB; ++j) {
        const int row = (wave * IB + j) * RPI + rsub;
        const int n = n0 + row;
        b_voff[j] = n < d.Cout ? (unsigned)(n * Kpad + 16 * (pos ^ f8_swz(row))) : 0x80000000u;
    }
    const bool fast = (d.Cin % F8_BK) == 0;                  // a K-step lies inside one filter tap
    int f_r = 0, f_s = 0, f_c = 0;
    const unsigned lds0 = lds_addr(lds);
    auto dma_step = [&](int ks, int buf) {
        const unsigned A = lds0 + (unsigned)((buf * STEP + (wave_u * IA) * RPI * RF) * 4);
        const unsigned B = lds0 + (unsigned)((buf * STEP + BM * RF + (wave_u * IB) * RPI * RF) * 4);
#pragma unroll
        for (int j = 0; j < IB; ++j) dma16(rs_b, B + j * (RPI * RF * 4), b_voff[j], (unsigned)(ks * F8_BK));
        if (fast) {
            if (f_c == 0) {
                const bool tap_ok = f_r < d.kh;
                const int hoff = f_r * d.b, woff = f_s * d.b;
#pragma unroll
                for (int j = 0; j < IA; ++j) {
                    const int nh = a_h[j] + hoff, nw = a_w[j] + woff;
                    const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                    const bool ok = tap_ok & ((nh | nw) >= 0) & (((nh | nw) & dmask) == 0) & (ih < d.Hi) & (iw < d.Wi);
                    a_voff[j] = ok ? (unsigned)(a_img[j] + (ih * d.Wi + iw) * d.Cin + a_c[j]) : 0x80000000u;
                }
            }
#pragma unroll
            for (int j = 0; j < IA; ++j) dma16(rs_a, A + j * (RPI * RF * 4), a_voff[j], (unsigned)f_c);
            f_c += F8_BK;
            if (f_c >= d.Cin) { f_c = 0; if (++f_s == d.kw) { f_s = 0; ++f_r; } }
        } else {
#pragma unroll
            for (int j = 0; j < IA; ++j) {                   // Cin % 16 == 0: a 16-byte chunk stays inside one tap
                const int k = ks * F8_BK + a_c[j];
                const int tap = k / d.Cin;
                const int c0 = k - tap * d.Cin;
                const int r = tap / d.kw, s_ = tap - r * d.kw;
                const int nh = a_h[j] + r * d.b, nw = a_w[j] + s_ * d.b;
                const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                const bool ok = (r < d.kh) & ((nh | nw) >= 0) & (((nh | nw) & dmask) == 0) & (ih < d.Hi) & (iw < d.Wi);
                dma16(rs_a, A + j * (RPI * RF * 4), ok ? (unsigned)(a_img[j] + (ih * d.Wi + iw) * d.Cin + c0) : 0x80000000u, 0u);
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
    // fragment addresses (words within a buffer): row = lane & 31 of each 32-row MFMA tile, logical chunks 2h and 2h + 1 (h = lane >> 5:
    // this lane's 32 consecutive K values), at their swizzled positions
    int fa[2][2], fb[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ra = wm * 64 + t * 32 + (lane & 31), rb = wn * 64 + t * 32 + (lane & 31);
            const int ch = 2 * (lane >> 5) + c;
            fa[t][c] = ra * RF + 4 * (ch ^ f8_swz(ra));
            fb[t][c] = BM * RF + rb * RF + 4 * (ch ^ f8_swz(rb));
        }
    const int one = 0x7F7F7F7F;                              // E8M0 block scales: 2^0 in every byte
    auto multiply = [&](int buf) {
        const float *S = lds + buf * STEP;
        i32x8 a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const i32x4 a0 = *reinterpret_cast<const i32x4 *>(S + fa[t][0]), a1 = *reinterpret_cast<const i32x4 *>(S + fa[t][1]);
            const i32x4 b0 = *reinterpret_cast<const i32x4 *>(S + fb[t][0]), b1 = *reinterpret_cast<const i32x4 *>(S + fb[t][1]);
            a[t] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            b[t] = i32x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[tm], b[tn], acc[tm][tn], 0, 0, 0, one, 0, one);
    };

    // K loop over a ring of three LDS buffers, counted waits (conv_bf16.hip: same structure)
    constexpr int NLD = IA + IB;
    auto wait_keep = [&](int steps_in_flight) {
        if (steps_in_flight <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    };
    const int pre = nks < NBUF - 1 ? nks : NBUF - 1;
    for (int s_ = 0; s_ < pre; ++s_) dma_step(s_, s_);
    wait_keep(pre - 1);
    __syncthreads();
    int rb = 0, wb = NBUF - 1;
    for (int ks = 0; ks < nks; ++ks) {
        if (ks + NBUF - 1 < nks) dma_step(ks + NBUF - 1, wb);
        multiply(rb);
        const int later = nks - 2 - ks;
        wait_keep(later < NBUF - 2 ? later : NBUF - 2);
        __syncthreads();
        rb = rb == NBUF - 1 ? 0 : rb + 1;
        wb = wb == NBUF - 1 ? 0 : wb + 1;
    }
    rn_wait_dma();

    // ---- epilogue through LDS (two passes of 64 rows): a lane owns CH consecutive channels of one output pixel: 16 for an fp8
    // result (16-byte stores, 16-byte addend loads), 4 for an fp32 result
    float *T = lds;
    constexpr int CH = YF32 ? 4 : 16;
    constexpr int CPR = BN / CH, RPP = 256 / CPR;
    const int cc = tid % CPR;
    const int col = n0 + CH * cc;
    const bool col_ok = col < d.Cout;
    unsigned char *yq = reinterpret_cast<unsigned char *>(yv);
    float *yf = reinterpret_cast<float *>(yv);
    // the lane's CH scale / shift values once per tile (they were re-read per row and element), and a dense result without the pixel
    // decomposition (round 4: the layers this kernel keeps -- 3x3 with <= 64 output channels, the fp32 head outputs -- are all epilogue)
    float scv[CH], shv[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        scv[j] = (col_ok && scale != nullptr) ? scale[col + j] : 1.f;
        shv[j] = (col_ok && shift != nullptr) ? shift[col + j] : 0.f;
    }
    const bool dense = d.os == 1 && d.oo_h == 0 && d.oo_w == 0 && d.Hy == d.Ho && d.Wy == d.Wo && d.y_batch_stride == (int64_t)HoWo * d.Cout &&
                       d.add_mode != 2 && (d.add_mode == 0 || d.add_batch_stride == d.y_batch_stride);
#pragma unroll 1
    for (int pass = 0; pass < BM / RP; ++pass) {
        if (pass) __syncthreads();
        if (wm == pass) {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        T[(tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * LDT + wn * 64 + tn * 32 + (lane & 31)] = acc[tm][tn][e];
        }
        __syncthreads();
        if (!col_ok) continue;
#pragma unroll 1
        for (int r = tid / CPR; r < RP; r += RPP) {
            const int64_t m = (int64_t)m0 + pass * RP + r;
            if (m >= M) break;
            int64_t off, aoff = -1;
            if (dense) {
                off = m * d.Cout + col;
                if (d.add_mode == 1) aoff = off;
            } else {
                const unsigned mu = (unsigned)m;
                const int n = (int)(mu / (unsigned)HoWo);
                const int rem = (int)(mu - (unsigned)n * (unsigned)HoWo);
                const int oh = (int)((unsigned)rem / (unsigned)d.Wo), ow = rem - oh * d.Wo;
                const int ph = oh * d.os + d.oo_h, pw = ow * d.os + d.oo_w;
                const int64_t pix = (int64_t)ph * d.Wy + pw;
                off = (int64_t)n * d.y_batch_stride + pix * d.Cout + col;
                if (d.add_mode == 1) aoff = (int64_t)n * d.add_batch_stride + pix * d.Cout + col;
                else if (d.add_mode == 2) aoff = (int64_t)n * d.add_batch_stride + ((int64_t)(oh >> 1) * d.Wa + (ow >> 1)) * d.Cout + col;
            }
            float v[CH];
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                const float4 t = *reinterpret_cast<const float4 *>(T + r * LDT + CH * cc + 4 * q);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
            int aq[CH / 4];
            if (aoff >= 0) {
#pragma unroll
                for (int q = 0; q < CH / 4; ++q) aq[q] = reinterpret_cast<const int *>(add + aoff)[q];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float u = v[j] * scv[j] + shv[j];
                if (aoff >= 0) {
                    const int word = aq[j >> 2];
                    const float a = (j & 3) == 0 ? __builtin_amdgcn_cvt_f32_fp8(word, 0) : (j & 3) == 1 ? __builtin_amdgcn_cvt_f32_fp8(word, 1)
                                  : (j & 3) == 2 ? __builtin_amdgcn_cvt_f32_fp8(word, 2) : __builtin_amdgcn_cvt_f32_fp8(word, 3);
                    u += a * fa_.add_scale;
                }
                if (d.act == 1) u = fmaxf(u, 0.f);
                else if (d.act == 2) u = 1.0f / (1.0f + expf(-u));
                v[j] = u;
            }
            if constexpr (YF32) {
                *reinterpret_cast<float4 *>(yf + off) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                const float is = fa_.out_inv_scale;
                i32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = f8_pack4(v[4 * q] * is, v[4 * q + 1] * is, v[4 * q + 2] * is, v[4 * q + 3] * is);
                *reinterpret_cast<i32x4 *>(yq + off) = o;
            }
        }
    }
}

template <bool YF32>
__global__ __launch_bounds__(256, 3) void conv_igemm_fp8_kernel(const rn_conv_desc d, const unsigned char *__restrict__ x,
                                                                const unsigned char *__restrict__ w, void *__restrict__ yv,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                const unsigned char *__restrict__ add, const Fp8Args fa_) {
    conv_igemm_fp8_tile<YF32>(d, x, w, yv, scale, shift, add, fa_, xcd_remap(blockIdx.x, gridDim.x));
}

// Grouped launch (rn_conv_igemm_grouped's form): the pyramid levels of a head layer, same weights and scalars, as ONE grid.
// The group's pointers are e4m3 bytes behind the float-typed fields of rn_conv_group.
template <bool YF32>
__global__ __launch_bounds__(256, 3) void conv_igemm_fp8_grouped_kernel(const rn_conv_group g, const unsigned char *__restrict__ w,
                                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                                        const Fp8Args fa_) {
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int i = 0; i < RN_MAX_GROUP - 1; ++i) p += (i + 1 < g.n && tile >= g.tile_end[i]) ? 1 : 0;
    rn_conv_desc d = g.d[0];
    const float *x = g.x[0], *add = g.add[0];
    float *y = g.y[0];
    int first = 0;
#pragma unroll
    for (int i = 1; i < RN_MAX_GROUP; ++i)
        if (p == i) { d = g.d[i]; x = g.x[i]; y = g.y[i]; add = g.add[i]; first = g.tile_end[i - 1]; }
    conv_igemm_fp8_tile<YF32>(d, reinterpret_cast<const unsigned char *>(x), w, y, scale, shift,
                              reinterpret_cast<const unsigned char *>(add), fa_, tile - first);
}

// The eight-wave 256 x 256 x 128 tile with the phased K loop (conv_fp8_p8.hip) for stride-1 same-size layers.
bool rn_fp8_p8_legal(const rn_conv_desc *d, int y_is_f32);
bool rn_fp8_p8_group_ok(const rn_conv_desc *d);
int rn_fp8_p8_launch(const rn_conv_desc *d, const void *x, const void *w, void *y, const float *scale, const float *shift, const void *add,
                     float add_scale, float out_inv_scale, hipStream_t stream);
int rn_fp8_p8_launch_grouped(const rn_conv_group *g, int tiles, const void *w, const float *scale, const float *shift, float add_scale,
                             float out_inv_scale, hipStream_t stream);
// RN_OPT_FP8_P8: 0 never, 2 wherever legal, 1 (default): wherever legal but the 3x3 layers with at most 64 output channels -- on BASELINE
// configs[4]'s layers it is faster or equal on every other shape it can compute, the short reductions and partial channel tiles included (profiles/r04_fp8_p8_by_shape.txt: 1x1 256 -> 1024 2.4x,
// 3x3 256 -> 256 1.6x, the whole forward pass 35.6 -> 24.5 ms).  RN_FP8_P8_MIN_K / RN_FP8_P8_MIN_TILES restrict mode 1 for A/B runs.
static inline bool fp8_p8_pick(const rn_conv_desc *d, int y_is_f32, int64_t tiles_in_launch) {
    const int mode = rn_get_option(RN_OPT_FP8_P8);
    if (mode == 0 || !rn_fp8_p8_legal(d, y_is_f32)) return false;
    if (mode == 2) return true;
    static const int min_k = [] { const char *e = getenv("RN_FP8_P8_MIN_K"); return e ? atoi(e) : 0; }();
    static const int min_tiles = [] { const char *e = getenv("RN_FP8_P8_MIN_TILES"); return e ? atoi(e) : 0; }();
    if (d->Cout <= 64 && d->kh * d->kw > 1) return false;      // a quarter of the 256-channel tile on a long reduction: 3x3 64 -> 64 measured 1.25 -> 1.54 ms
    return d->kh * d->kw * d->Cin >= min_k && tiles_in_launch >= min_tiles;
}
static inline bool fp8_group_is_p8(const rn_conv_group *g, int y_is_f32) {
    int64_t t = 0;
    for (int i = 0; i < g->n; ++i) t += (((int64_t)g->d[i].N * g->d[i].Ho * g->d[i].Wo + 255) / 256) * ((g->d[i].Cout + 255) / 256);
    for (int i = 0; i < g->n; ++i)
        if (!rn_fp8_p8_group_ok(&g->d[i]) || !fp8_p8_pick(&g->d[i], y_is_f32, t)) return false;
    return true;
}
// Tile shape rn_conv_igemm_fp8_grouped will use for this group (rows * 1000 + cols): the caller builds tile_end with it.
extern "C" int rn_conv_igemm_fp8_tile_rows(const rn_conv_group *g, int y_is_f32) {
    if (g->n < 1 || g->n > RN_MAX_GROUP) return 0;
    return fp8_group_is_p8(g, y_is_f32) ? 256 * 1000 + 256 : 128 * 1000 + 128;
}

// The tile a SINGLE launch (rn_conv_igemm_fp8) takes for this problem, rows * 1000 + cols (profiling / tests; the grouped form has fewer instances).
extern "C" int rn_conv_igemm_fp8_tile(const rn_conv_desc *d, int y_is_f32) {
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    return fp8_p8_pick(d, y_is_f32, ((M + 255) / 256) * ((d->Cout + 255) / 256)) ? 256 * 1000 + 256 : 128 * 1000 + 128;
}

static int check_desc_fp8(const rn_conv_desc *d, int y_is_f32) {
    if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return RN_EINVAL;
    if (d->Cin < 16 || (d->Cin & 15) || (d->Cout & (y_is_f32 ? 3 : 15)) || d->w_format != 0) return RN_EINVAL;
    if ((int64_t)d->Hi * d->Wi * d->Cin > 0x7fffffffLL) return RN_EINVAL;
    const int64_t HoWo = (int64_t)d->Ho * d->Wo, span = 127 / HoWo + 2;
    if (d->x_batch_stride < 0 || (span - 1) * d->x_batch_stride + (int64_t)d->Hi * d->Wi * d->Cin > 0x7fffffffLL) return RN_EINVAL;
    const int64_t Kpad = ((int64_t)d->kh * d->kw * d->Cin + 63) / 64 * 64;
    if (d->Cout * Kpad > 0x7fffffffLL || (int64_t)d->N * HoWo > 0x7fffffffLL) return RN_EINVAL;
    if (d->kh <= 0 || d->kw <= 0 || d->div_shift < 0 || d->div_shift > 2) return RN_EINVAL;
    if (d->add_mode < 0 || d->add_mode > 2 || d->act < 0 || d->act > 2) return RN_EINVAL;
    if (d->mask_mode != 0 || d->in_relu || d->add2_mode != 0 || d->w_batch_stride != 0) return RN_EINVAL;   // forward form only
    if (d->os < 1 || d->oo_h < 0 || d->oo_w < 0) return RN_EINVAL;
    if ((d->Ho - 1) * d->os + d->oo_h >= d->Hy || (d->Wo - 1) * d->os + d->oo_w >= d->Wy) return RN_EINVAL;
    if (d->os != 1 && d->add_mode == 2) return RN_EINVAL;
    if (!y_is_f32 && ((d->y_batch_stride & 15) || (d->add_batch_stride & 3))) return RN_EINVAL;
    return RN_OK;
}

extern "C" int rn_conv_igemm_fp8(const rn_conv_desc *d, const void *x_q, const void *w_q, void *y, int y_is_f32, const float *scale,
                                 const float *shift, const void *add_q, float add_scale, float out_inv_scale, void *stream) {
    const int rc = check_desc_fp8(d, y_is_f32);
    if (rc) return rc;
    if ((d->add_mode != 0) != (add_q != nullptr)) return RN_EINVAL;
    if (((uintptr_t)x_q & 15) || ((uintptr_t)w_q & 15) || ((uintptr_t)y & 15) || ((uintptr_t)add_q & 3)) return RN_EINVAL;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    if (!((uintptr_t)add_q & 15) && fp8_p8_pick(d, y_is_f32, ((M + 255) / 256) * ((d->Cout + 255) / 256)))
        return rn_fp8_p8_launch(d, x_q, w_q, y, scale, shift, add_q, add_scale, out_inv_scale, (hipStream_t)stream);
    const int64_t tiles = ((M + 127) / 128) * ((d->Cout + 127) / 128);
    if (tiles > 0x7fffffff) return RN_EINVAL;
    Fp8Args a;
    a.add_scale = add_scale;
    a.out_inv_scale = out_inv_scale;
    const dim3 grid((unsigned)tiles), block(256);
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(x_q), *wb = reinterpret_cast<const unsigned char *>(w_q);
    const unsigned char *ab = reinterpret_cast<const unsigned char *>(add_q);
    if (y_is_f32) hipLaunchKernelGGL((conv_igemm_fp8_kernel<true>), grid, block, 0, (hipStream_t)stream, *d, xb, wb, y, scale, shift, ab, a);
    else hipLaunchKernelGGL((conv_igemm_fp8_kernel<false>), grid, block, 0, (hipStream_t)stream, *d, xb, wb, y, scale, shift, ab, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// One launch for up to RN_MAX_GROUP problems that share weights, scales and activation (tile_end[i] = running sum of
// ceil(N*Ho*Wo / 128) * ceil(Cout / 128)); x / y / add of the group are e4m3 (y: or fp32) behind the float-typed fields.
extern "C" int rn_conv_igemm_fp8_grouped(const rn_conv_group *g, const void *w_q, int y_is_f32, const float *scale, const float *shift,
                                         float add_scale, float out_inv_scale, void *stream) {
    if (g->n < 1 || g->n > RN_MAX_GROUP || ((uintptr_t)w_q & 15)) return RN_EINVAL;
    const rn_conv_desc &d0 = g->d[0];
    const bool p8 = fp8_group_is_p8(g, y_is_f32);               // the caller's tile_end must follow rn_conv_igemm_fp8_tile_rows()
    const int TR = p8 ? 256 : 128;
    int prev = 0;
    for (int i = 0; i < g->n; ++i) {
        const rn_conv_desc &d = g->d[i];
        const int rc = check_desc_fp8(&d, y_is_f32);
        if (rc) return rc;
        if (d.Cin != d0.Cin || d.Cout != d0.Cout || d.kh != d0.kh || d.kw != d0.kw || d.act != d0.act) return RN_EINVAL;
        if ((d.add_mode != 0) != (g->add[i] != nullptr)) return RN_EINVAL;
        if (((uintptr_t)g->x[i] & 15) || ((uintptr_t)g->y[i] & 15) || ((uintptr_t)g->add[i] & (p8 ? 15 : 3))) return RN_EINVAL;
        const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
        const int64_t tiles = ((M + TR - 1) / TR) * ((d.Cout + TR - 1) / TR);
        if (g->tile_end[i] - prev != tiles) return RN_EINVAL;
        prev = g->tile_end[i];
    }
    if (p8) return rn_fp8_p8_launch_grouped(g, prev, w_q, scale, shift, add_scale, out_inv_scale, (hipStream_t)stream);
    Fp8Args a;
    a.add_scale = add_scale;
    a.out_inv_scale = out_inv_scale;
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(w_q);
    if (y_is_f32) hipLaunchKernelGGL((conv_igemm_fp8_grouped_kernel<true>), dim3((unsigned)prev), dim3(256), 0, (hipStream_t)stream, *g, wb, scale, shift, a);
    else hipLaunchKernelGGL((conv_igemm_fp8_grouped_kernel<false>), dim3((unsigned)prev), dim3(256), 0, (hipStream_t)stream, *g, wb, scale, shift, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
