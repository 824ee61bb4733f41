// bf16 implicit-GEMM convolution and weight gradient on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16, fp32
// accumulation; dense peak ~2.5 PFLOP/s) -- the reduced-precision form of conv_igemm.hip / conv_wgrad.hip for BASELINE
// configs[2] ("bf16 MFMA").  Same convolutions of the reference (every nn.Conv2d of D/model.py with its fused frozen
// batch-norm / bias / residual / ReLU / sigmoid / FPN upsample-add epilogue, D/model.py:59-205, D/utils.py:12-80), same
// rn_conv_desc geometry (so the same call computes data gradients), different storage: activations and packed weights
// are bf16 (NHWC / [Cout][kh][kw][Cin]), accumulation and the whole epilogue are fp32, the result is rounded once on
// the way out (bf16, or fp32 where the consumer is the loss).  Parameters stay fp32 master copies: the packed bf16
// weights are derived per step (rn_pack_weights, then rn_f32_to_bf16).
//
// Tile / staging: identical to the fp32 kernel BYTE for byte -- a staged row is 64 bytes of K (32 bf16 instead of 16
// floats), rows go global -> LDS by direct-to-LDS buffer loads, out-of-range offsets are the zero padding, the 16-byte
// chunks of a row are XOR-permuted on the load's source address and on the fragment read.  A ds_read_b128 at
// [row = lane&31][chunk 2*st + (lane>>5)] is exactly the A / B operand of one 32x32x16 MFMA (8 consecutive k per lane,
// k 0-7 in lanes 0-31, 8-15 in lanes 32-63), so per 64-byte K-step a wave issues 8 b128 reads against 8 MFMAs of 32
// cycles -- where the fp32 kernel runs 32 MFMAs of 64 cycles on the same bytes.  That ratio is the point: per byte
// staged the matrix cores are busy 1/8 as long, so this kernel lives on L2 / LDS / HBM bandwidth, not on MFMA issue
// (128x128 tile: 64 FLOP per staged byte).
//
// Weight gradient: dW[co][k] = sum over pixels of dY[pixel][co] * X[pixel + tap][ci] has the REDUCTION index (pixels) as
// the row of both NHWC operands, while a 32x32x16 operand wants 8 consecutive reduction elements per lane.  The tiles
// are staged as they lie ([pixel][channel], direct-to-LDS like conv_wgrad.hip) and read through ds_read_b64_tr_b16,
// gfx950's transposing LDS read: per 16 lanes a 4-pixel x 16-channel block comes back channel-major, two of them make
// one operand.  fp32 atomics into the same packed fp32 [Cout][Kpad] gradient buffer the fp32 path uses.
//
// Roofline: MFMA by FLOPs (2.5 PF dense), in practice the staging bandwidth (see above) and, for the 1x1 layers, HBM.
#include <stdlib.h>

#include "common.h"
#include "conv_wgrad_geom.h"

#ifndef RN_WG_KO
#define RN_WG_KO 0       // conv_wgrad.hip: 1 no tile atomics, 2 workgroup-scope atomics (RN_EXPERIMENT builds, timing only)
#endif
#if RN_WG_KO == 1
#define RN_WG_ATOMIC(P, V) do { if ((V) == 1.2345e-30f) atomicAdd((P), (V)); } while (0)
#elif RN_WG_KO == 2
#define RN_WG_ATOMIC(P, V) __hip_atomic_fetch_add((P), (V), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define RN_WG_ATOMIC(P, V) atomicAdd((P), (V))
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BF_BK 32                 // K elements per step = 64 bytes per staged row
#ifndef BF_NBUF
#define BF_NBUF 2                // LDS ring depth of the implicit-GEMM kernel (K-steps staged or in flight per workgroup)
#endif
#ifndef BF_OCC
#define BF_OCC 4                 // waves per SIMD the register budget is cut for = workgroups per CU that fit next to the LDS
#endif

__device__ __forceinline__ int bf_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
__device__ __forceinline__ int bf_swz(int row) { return (row >> 2) & 3; }     // 64-byte rows (conv_igemm_tile.h: lds_swz<16>)

__device__ __forceinline__ void bf_load4(const __bf16 *p, float (&v)[4]) {
    const bf16x4 q = *reinterpret_cast<const bf16x4 *>(p);
    v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
}

// ---------------------------------------------------------------------------------------------- casts
__global__ void f32_to_bf16_kernel(const float *__restrict__ src, __bf16 *__restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4 *>(src + i);
        bf16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;     // plain casts: RNE, a NaN stays a NaN
        *reinterpret_cast<bf16x4 *>(dst + i) = o;
    } else {
        for (int64_t k = i; k < n; ++k) dst[k] = (__bf16)src[k];
    }
}
__global__ void bf16_to_f32_kernel(const __bf16 *__restrict__ src, float *__restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float v[4];
        bf_load4(src + i, v);
        *reinterpret_cast<float4 *>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int64_t k = i; k < n; ++k) dst[k] = (float)src[k];
    }
}
extern "C" int rn_f32_to_bf16(const float *src, void *dst, int64_t n, void *stream) {
    if (n <= 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return RN_EINVAL;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(rn_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<__bf16 *>(dst), n);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
extern "C" int rn_bf16_to_f32(const void *src, float *dst, int64_t n, void *stream) {
    if (n <= 0 || ((uintptr_t)src & 7) || ((uintptr_t)dst & 15)) return RN_EINVAL;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(rn_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16 *>(src), dst, n);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ---------------------------------------------------------------------------------------------- implicit GEMM
// One (64*WM) x (64*WN) output tile by WM x WN waves, each a 64 x 64 sub-tile = 2 x 2 accumulators of the 32x32x16 MFMA:
//   <2,2>  128 x 128, 256 threads, four workgroups per CU  -- 64 FLOP per staged byte;
//   <4,4>  256 x 256, 1024 threads, one workgroup per CU   -- 128 FLOP per staged byte: the per-wave work (8 MFMAs and
//          8 ds_read_b128 per K-step) is the same, only HALF the bytes come through L2 -> LDS per FLOP, which is what
//          bounds the small tile (~800 TFLOP/s = 12.5 TB/s of staging against 17-19 TB/s the LDS-DMA path delivers from L2).
//          Used where the problem has enough 256-tiles to fill the chip (launcher).
// YF32: the output is written as fp32 (head outputs feeding the loss) instead of bf16; addend and mask are bf16.
template <bool YF32, int WM, int WN, bool DENSE, int TM = 2>
__device__ __forceinline__ void conv_igemm_bf16_tile(const rn_conv_desc &d, const __bf16 *__restrict__ x,
                                                     const __bf16 *__restrict__ w, void *__restrict__ yv,
                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                     const __bf16 *__restrict__ add, const __bf16 *__restrict__ mask, const int tile) {
    constexpr int TN = 2;                                    // 32-wide MFMA tiles per wave along N; TM along M (2: 64 rows, 4: 128 rows)
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN, NT = 64 * NW;
    constexpr int RF = 16;                                   // floats (4-byte words) per staged row: 64 bytes
    constexpr int RPI = 16;                                  // rows one wave instruction fills (1 KiB / 64 B)
    constexpr int IA = BM / RPI / NW, IB = BN / RPI / NW;    // DMA instructions per wave per K-step and operand
    constexpr int STEP = (BM + BN) * RF;                     // 4-byte words per buffer: A rows, then B rows
    constexpr int LDT = BN + 4;                              // epilogue: padded fp32 output tile row
    constexpr int RP = NW == 4 ? 64 : 32;                    // tile rows per epilogue pass (a multiple of an MFMA tile's 32)
    constexpr int NBUF = NW == 4 ? BF_NBUF : 3;              // the one-workgroup-per-CU tile keeps a third K-step in flight (+7 % measured)
    static_assert(IA >= 1 && IB >= 1 && BM % RP == 0, "tile shape");
    constexpr int LDSF = NBUF * STEP > RP * LDT ? NBUF * STEP : RP * LDT;
    __shared__ float lds[LDSF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ntn = (d.Cout + BN - 1) / BN;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int HoWo = d.Ho * d.Wo;
    const int64_t M = (int64_t)d.N * HoWo;
    const int K = d.kh * d.kw * d.Cin;
    const int Kpad = (K + 31) / 32 * 32;                     // packed weight rows: zero-padded to a multiple of 32 elements
    const int nks = Kpad / BF_BK;
    const int dmask = (1 << d.div_shift) - 1;

    // buffer descriptors (scalar); out-of-range offsets read as zero = padding / rows past M / weight rows past Cout
    const int n_first = (int)(m0 / HoWo);
    const int64_t x_elems = ((int64_t)d.N - 1 - n_first) * d.x_batch_stride + (int64_t)d.Hi * d.Wi * d.Cin;
    const v4i32 rs_a = make_rsrc(x + (int64_t)n_first * d.x_batch_stride,
                                 (unsigned)(x_elems * 2 > 0x7FFFFFFF ? 0x7FFFFFFF : x_elems * 2));
    const v4i32 rs_b = make_rsrc(w, (unsigned)((int64_t)d.Cout * Kpad * 2));

    // per-lane staging geometry: instruction j of this wave fills rows (wave*I + j)*16 .. +15 of the operand; the lane
    // fills 16-byte position pos of row rsub of them, i.e. fetches logical chunk pos ^ swz(row) = 8 elements of K
    const int pos = lane & 3, rsub = lane >> 2;
    int a_h[IA], a_w[IA], a_img[IA], a_c[IA];
    unsigned a_voff[IA];
    const int rel0 = m0 - n_first * HoWo;
#pragma unroll
    for (int j = 0; j < IA; ++j) {
        const int row = (wave * IA + j) * RPI + rsub;
        a_c[j] = 8 * (pos ^ bf_swz(row));
        if ((int64_t)m0 + row < M) {
            const unsigned rel = (unsigned)(rel0 + row);
            const unsigned n = rel / (unsigned)HoWo;
            const unsigned rem = rel - n * (unsigned)HoWo;
            const unsigned oh = rem / (unsigned)d.Wo, ow = rem - oh * (unsigned)d.Wo;
            a_img[j] = (int)((int64_t)n * d.x_batch_stride * 2);
            a_h[j] = (int)oh * d.a + d.p;
            a_w[j] = (int)ow * d.a + d.p_w;
        } else {
            a_img[j] = 0;
            a_h[j] = -(1 << 28);                             // fails every bounds test
            a_w[j] = 0;
        }
        a_voff[j] = 0x80000000u;
    }
    unsigned b_voff[IB];
#pragma unroll
    for (int j = 0; j < IB; ++j) {
        const int row = (wave * IB + j) * RPI + rsub;
        const int n = n0 + row;
        b_voff[j] = n < d.Cout ? (unsigned)((n * Kpad + 8 * (pos ^ bf_swz(row))) * 2) : 0x80000000u;
    }

    // fast path: Cin a multiple of the K-step -> a step lies inside one filter tap; offsets change only with the tap
    const bool fast = (d.Cin % BF_BK) == 0;
    int f_r = 0, f_s = 0, f_c = 0;
    const unsigned lds0 = lds_addr(lds);
    auto dma_step = [&](int ks, int buf) {
        const unsigned A = lds0 + (unsigned)((buf * STEP + (wave_u * IA) * RPI * RF) * 4);
        const unsigned B = lds0 + (unsigned)((buf * STEP + BM * RF + (wave_u * IB) * RPI * RF) * 4);
#pragma unroll
        for (int j = 0; j < IB; ++j) dma16(rs_b, B + j * (RPI * RF * 4), b_voff[j], (unsigned)(ks * BF_BK * 2));
        if (fast) {
            if (f_c == 0) {                                  // new tap (wave-uniform)
                const bool tap_ok = f_r < d.kh;
                const int hoff = f_r * d.b, woff = f_s * d.b;
#pragma unroll
                for (int j = 0; j < IA; ++j) {
                    const int nh = a_h[j] + hoff, nw = a_w[j] + woff;
                    const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                    const bool ok = tap_ok & ((nh | nw) >= 0) & (((nh | nw) & dmask) == 0) & (ih < d.Hi) & (iw < d.Wi);
                    a_voff[j] = ok ? (unsigned)(a_img[j] + ((ih * d.Wi + iw) * d.Cin + a_c[j]) * 2) : 0x80000000u;
                }
            }
#pragma unroll
            for (int j = 0; j < IA; ++j) dma16(rs_a, A + j * (RPI * RF * 4), a_voff[j], (unsigned)(f_c * 2));
            f_c += BF_BK;
            if (f_c >= d.Cin) { f_c = 0; if (++f_s == d.kw) { f_s = 0; ++f_r; } }
        } else {
#pragma unroll
            for (int j = 0; j < IA; ++j) {                   // Cin % 8 == 0: a 16-byte chunk stays inside one tap
                const int k = ks * BF_BK + a_c[j];
                const int tap = k / d.Cin;
                const int c0 = k - tap * d.Cin;
                const int r = tap / d.kw, s_ = tap - r * d.kw;
                const int nh = a_h[j] + r * d.b, nw = a_w[j] + s_ * d.b;
                const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                const bool ok = (r < d.kh) & ((nh | nw) >= 0) & (((nh | nw) & dmask) == 0) & (ih < d.Hi) & (iw < d.Wi);
                dma16(rs_a, A + j * (RPI * RF * 4), ok ? (unsigned)(a_img[j] + ((ih * d.Wi + iw) * d.Cin + c0) * 2) : 0x80000000u, 0u);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses (4-byte words within a buffer): row = lane & 31 of each 32-row MFMA tile, logical chunk
    // 2*st + (lane >> 5) of sub-step st, at its swizzled position
    int fa[TM][2], fb[TN][2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int ra = wm * (32 * TM) + t * 32 + (lane & 31);
            fa[t][st] = ra * RF + 4 * ((2 * st + (lane >> 5)) ^ bf_swz(ra));
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int rb = wn * (32 * TN) + t * 32 + (lane & 31);
            fb[t][st] = BM * RF + rb * RF + 4 * ((2 * st + (lane >> 5)) ^ bf_swz(rb));
        }
    }
    auto multiply = [&](int buf) {
        const float *S = lds + buf * STEP;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = *reinterpret_cast<const bf16x8 *>(S + fa[t][st]);
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const bf16x8 *>(S + fb[t][st]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    };

    // K loop over a ring of NBUF LDS buffers: the loads of step ks + NBUF-1 are issued before the MFMAs of step ks, one
    // barrier per step; the wait in front of the barrier is COUNTED -- everything but the steps issued after ks+1 must have
    // landed -- so with NBUF > 2 loads stay in flight across barriers.  (A K-step is 8 MFMAs of 32 cycles per wave here
    // against 32 of 64 in the fp32 kernel: one step of prefetch no longer covers an L2 round trip.)
    constexpr int NLD = IA + IB;                             // loads one wave issues per K-step
    auto wait_keep = [&](int steps_in_flight) {              // wave-uniform; vmcnt takes an immediate
        if (steps_in_flight <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (steps_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else if (steps_in_flight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLD) : "memory");
    };
    static_assert(NBUF >= 2 && NBUF <= 5, "ring depth");
    const int pre = nks < NBUF - 1 ? nks : NBUF - 1;
    for (int s_ = 0; s_ < pre; ++s_) dma_step(s_, s_);
    wait_keep(pre - 1);
    __syncthreads();
#ifndef BF_ABL
#define BF_ABL 0                 // knock-outs for profiles/: 1 = no epilogue, 2 = no MFMAs, 3 = no staging loads after the prologue
#endif
    int rb = 0, wb = NBUF - 1;
    for (int ks = 0; ks < nks; ++ks) {
        if (BF_ABL != 3 && ks + NBUF - 1 < nks) dma_step(ks + NBUF - 1, wb);
        if (BF_ABL != 2) multiply(rb);
        const int later = nks - 2 - ks;                      // steps issued after ks+1 that may stay in flight
        wait_keep(later < NBUF - 2 ? later : NBUF - 2);
        __syncthreads();
        rb = rb == NBUF - 1 ? 0 : rb + 1;
        wb = wb == NBUF - 1 ? 0 : wb + 1;
    }
    rn_wait_dma();
    if (BF_ABL == 1) {                                       // keep the accumulators alive, store next to nothing
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) t += acc[i][j][0] + acc[i][j][15];
        if (t == 1.2345e-30f) reinterpret_cast<float *>(yv)[0] = t;
        return;
    }

    // ---- epilogue: v = scale[c]*acc + shift[c]; [mask before add]; v += add; act; [mask after]; one rounding on the store.
    // The accumulator tile goes through LDS (two passes of 64 rows) so that global memory sees whole row segments in
    // 16-byte accesses: a lane owns CH consecutive channels of one output pixel -- 8 for a bf16 result (its addend and mask
    // are 16-byte loads too; 8-byte accesses run at 0.54-0.70 of the 16-byte rate), 4 for an fp32 result.
    float *T = lds;
    constexpr int CH = YF32 ? 4 : 8;
    constexpr int CPR = BN / CH, RPP = NT / CPR;             // chunks per tile row, rows per pass of stores
    const int cc = tid % CPR;
    const int col = n0 + CH * cc;
    const bool col_ok = col < d.Cout;                        // Cout % CH == 0 (checked by the launcher)
    float sc[CH], sh[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        sc[j] = (col_ok && scale != nullptr) ? scale[col + j] : 1.f;
        sh[j] = (col_ok && shift != nullptr) ? shift[col + j] : 0.f;
    }
    __bf16 *yb = reinterpret_cast<__bf16 *>(yv);
    float *yf = reinterpret_cast<float *>(yv);
    // DENSE (a template parameter: both paths in one kernel cost registers the K loop needs): result and addend in the problem's
    // own pixel order, so the offset of row m is m * Cout and there is nothing to decompose (bf16_desc_is_dense)
#pragma unroll 1
    for (int pass = 0; pass < BM / RP; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row0 = wm * (32 * TM) + tm * 32;       // this wave's 32-row MFMA tile: staged in the pass that holds it
            if (row0 / RP == pass) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        T[(row0 % RP + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * LDT + wn * (32 * TN) + tn * 32 + (lane & 31)] = acc[tm][tn][e];
            }
        }
        __syncthreads();
        if (!col_ok) continue;
        // The addend / mask operands of ALL rows this lane finishes in the pass are requested first (rows past M re-read
        // row M-1), then the rows are finished: one memory round trip per pass instead of one per row -- on the 1x1 layers
        // (a handful of K-steps per tile) the serialised round trips were most of the kernel (conv_igemm_tile.h, same fix).
        constexpr int NR = RP / RPP;                         // rows per lane per pass
        constexpr int G = NR % 2 == 0 ? 2 : 1;               // rows per group (more costs registers the K loop needs: spills)
        typedef __bf16 opv __attribute__((ext_vector_type(CH)));
        // the mask of CH consecutive elements: the bf16 tensor itself, or (mask_mode | RN_MASK_BITS) its sign bits (common.h) as 1 / 0
        const bool mbits = (d.mask_mode & RN_MASK_BITS) != 0;
        const int mmode = d.mask_mode & 3;
        auto load_mask = [&](const int64_t off) -> opv {
            if (!mbits) return *reinterpret_cast<const opv *>(mask + off);
            const unsigned b = rn_sign_bits(mask, off, CH);
            opv m;
#pragma unroll
            for (int j = 0; j < CH; ++j) m[j] = (__bf16)((b >> j) & 1u ? 1.0f : 0.0f);
            return m;
        };
#pragma unroll 1
        for (int g0 = 0; g0 < NR; g0 += G) {
            int64_t off_[G];
            opv mk_[G], ad_[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int64_t mr = (int64_t)m0 + pass * RP + tid / CPR + (g0 + i) * RPP;
                const int64_t m = mr < M ? mr : M - 1;
                if constexpr (DENSE) {
                    off_[i] = m * d.Cout + col;
                    if (d.mask_mode != 0) mk_[i] = load_mask(off_[i]);
                    if (d.add_mode == 1) ad_[i] = *reinterpret_cast<const opv *>(add + off_[i]);
                    continue;
                }
                const unsigned mu = (unsigned)m;               // m < 2^31 (launcher): 32-bit divisions (a 64-bit one is ~100 VALU instructions)
                const int n = (int)(mu / (unsigned)HoWo);
                const int rem = (int)(mu - (unsigned)n * (unsigned)HoWo);
                const int oh = (int)((unsigned)rem / (unsigned)d.Wo), ow = rem - oh * d.Wo;
                const int ph = oh * d.os + d.oo_h, pw = ow * d.os + d.oo_w;
                const int64_t pix = (int64_t)ph * d.Wy + pw;
                off_[i] = (int64_t)n * d.y_batch_stride + pix * d.Cout + col;
                if (d.mask_mode != 0) mk_[i] = load_mask(off_[i]);
                if (d.add_mode == 1) ad_[i] = *reinterpret_cast<const opv *>(add + (int64_t)n * d.add_batch_stride + pix * d.Cout + col);
                else if (d.add_mode == 2)                      // nearest x2 upsample, cropped (D/model.py:88-108)
                    ad_[i] = *reinterpret_cast<const opv *>(add + (int64_t)n * d.add_batch_stride + ((int64_t)(oh >> 1) * d.Wa + (ow >> 1)) * d.Cout + col);
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int r = tid / CPR + (g0 + i) * RPP;
                if ((int64_t)m0 + pass * RP + r < M) {
                    float v[CH];
#pragma unroll
                    for (int q = 0; q < CH / 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4 *>(T + r * LDT + CH * cc + 4 * q);
                        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        float u = v[j] * sc[j] + sh[j];
                        if (mmode == 1) u = (float)mk_[i][j] > 0.f ? u : 0.f;
                        if (d.add_mode != 0) u += (float)ad_[i][j];
                        if (d.act == 1) u = fmaxf(u, 0.f);
                        else if (d.act == 2) u = 1.0f / (1.0f + expf(-u));
                        if (mmode == 2) u = (float)mk_[i][j] > 0.f ? u : 0.f;
                        v[j] = u;
                    }
                    if constexpr (YF32) {
                        *reinterpret_cast<float4 *>(yf + off_[i]) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        bf16x8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
                        *reinterpret_cast<bf16x8 *>(yb + off_[i]) = o;
                        if (d.sign_out != nullptr) {           // the sign of what was STORED (after the rounding to bf16)
                            float st[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) st[j] = (float)o[j];
                            rn_sign_store8(reinterpret_cast<unsigned *>(d.sign_out), off_[i], st);
                        }
                    }
                }
            }
        }
    }
}

template <bool YF32, int WM, int WN, bool DENSE, int TM = 2>
__global__ __launch_bounds__(64 * WM * WN, TM == 4 ? 2 : BF_OCC) void conv_igemm_bf16_kernel(const rn_conv_desc d, const __bf16 *__restrict__ x,
                                                                 const __bf16 *__restrict__ w, void *__restrict__ yv,
                                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                                 const __bf16 *__restrict__ add, const __bf16 *__restrict__ mask) {
    conv_igemm_bf16_tile<YF32, WM, WN, DENSE, TM>(d, x, w, yv, scale, shift, add, mask, bf_xcd_remap(blockIdx.x, gridDim.x));
}

// Grouped launch (rn_conv_igemm_grouped's form): up to RN_MAX_GROUP problems sharing weights and epilogue scalars -- the five
// pyramid levels of a head layer -- as ONE grid; the workgroup looks up its problem by tile id (wave-uniform).
template <bool YF32, int WM, int WN, bool DENSE, int TM = 2>
__global__ __launch_bounds__(64 * WM * WN, TM == 4 ? 2 : BF_OCC) void conv_igemm_bf16_grouped_kernel(const rn_conv_group g, const __bf16 *__restrict__ w,
                                                                         const float *__restrict__ scale,
                                                                         const float *__restrict__ shift) {
    const int tile = bf_xcd_remap(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int i = 0; i < RN_MAX_GROUP - 1; ++i) p += (i + 1 < g.n && tile >= g.tile_end[i]) ? 1 : 0;
    rn_conv_desc d = g.d[0];
    const float *x = g.x[0], *add = g.add[0], *mask = g.mask[0];
    float *y = g.y[0];
    int first = 0;
#pragma unroll
    for (int i = 1; i < RN_MAX_GROUP; ++i)
        if (p == i) { d = g.d[i]; x = g.x[i]; y = g.y[i]; add = g.add[i]; mask = g.mask[i]; first = g.tile_end[i - 1]; }
    conv_igemm_bf16_tile<YF32, WM, WN, DENSE, TM>(d, reinterpret_cast<const __bf16 *>(x), w, y, scale, shift,
                                       reinterpret_cast<const __bf16 *>(add), reinterpret_cast<const __bf16 *>(mask), tile - first);
}

// The eight-wave 256 x 256 x 64 tile with the phased K loop (conv_bf16_p8.hip) for stride-1 same-size layers.
bool rn_bf16_p8_legal(const rn_conv_desc *d, int y_is_f32);
int rn_bf16_p8_launch(const rn_conv_desc *d, const void *x, const void *w, void *y, const float *scale, const float *shift,
                      const void *add, const void *mask, hipStream_t stream);
int rn_bf16_p8_launch_grouped(const rn_conv_group *g, int tiles, const void *w, const float *scale, const float *shift, hipStream_t stream);
// RN_OPT_BF16_P8: 0 never, 2 wherever legal, 1 (default): full 256-channel tiles (Cout % 256 == 0), at least 200 tiles in the launch
// (RN_BF16_P8_MIN_TILES; one workgroup per CU: below ~a round of the 256 CUs the 128 x 128 tile's four workgroups per CU win), and either a
// reduction of at least 256 (RN_BF16_P8_MIN_K) or NO mask operand: the short reductions are all epilogue, and the kernel requests every
// addend of a tile up front when there is no mask (the bottlenecks' conv3 + identity + ReLU: 1x1 64 -> 256 1.34 -> 0.96 ms per step, 128 ->
// 512 0.81 -> 0.60, 256 -> 1024 0.70 -> 0.53) and both operands of two pixel blocks at a time when there is one (their data gradients:
// equal to the 128 x 128 kernel at K = 64 / 128, 5 % faster at K = 256).  Per layer shape: profiles/r04_bf16_p8_by_shape.txt, r04_bf16_p8_short_k.txt.
static inline bool bf16_p8_pick(const rn_conv_desc *d, int y_is_f32, int64_t tiles_in_launch) {
    const int mode = rn_get_option(RN_OPT_BF16_P8);
    if (mode == 0 || !rn_bf16_p8_legal(d, y_is_f32)) return false;
    if (mode == 2) return true;
    static const int min_k = [] { const char *e = getenv("RN_BF16_P8_MIN_K"); return e ? atoi(e) : 256; }();
    static const int min_tiles = [] { const char *e = getenv("RN_BF16_P8_MIN_TILES"); return e ? atoi(e) : 200; }();
    if ((d->Cout & 255) != 0 || tiles_in_launch < min_tiles) return false;
    return d->kh * d->kw * d->Cin >= min_k || d->mask_mode == 0;
}
static inline bool bf16_group_is_p8(const rn_conv_group *g, int y_is_f32) {
    int64_t t = 0;
    for (int i = 0; i < g->n; ++i) t += (((int64_t)g->d[i].N * g->d[i].Ho * g->d[i].Wo + 255) / 256) * ((g->d[i].Cout + 255) / 256);
    for (int i = 0; i < g->n; ++i)
        if (!bf16_p8_pick(&g->d[i], y_is_f32, t)) return false;
    return true;
}

static inline bool bf16_desc_is_dense(const rn_conv_desc *d) {
    return d->os == 1 && d->oo_h == 0 && d->oo_w == 0 && d->Hy == d->Ho && d->Wy == d->Wo &&
           d->y_batch_stride == (int64_t)d->Ho * d->Wo * d->Cout && d->add_mode != 2 &&
           (d->add_mode == 0 || d->add_batch_stride == d->y_batch_stride);
}

static inline int check_desc_bf16(const rn_conv_desc *d) {
    if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return RN_EINVAL;
    if (d->Cin < 8 || (d->Cin & 7) || (d->Cout & 3) || d->w_format != 0) return RN_EINVAL;   // 16-byte chunks of 8 channels; 16-byte stores
    if ((int64_t)d->Hi * d->Wi * d->Cin * 2 > 0x7fffffffLL) return RN_EINVAL;
    const int64_t HoWo = (int64_t)d->Ho * d->Wo, span = 255 / HoWo + 2;    // images a 256-row tile can touch
    if (d->x_batch_stride < 0 || ((span - 1) * d->x_batch_stride + (int64_t)d->Hi * d->Wi * d->Cin) * 2 > 0x7fffffffLL) return RN_EINVAL;
    const int64_t Kpad = ((int64_t)d->kh * d->kw * d->Cin + 31) / 32 * 32;
    if (d->Cout * Kpad * 2 > 0x7fffffffLL || (int64_t)d->N * HoWo > 0x7fffffffLL) return RN_EINVAL;
    if (d->kh <= 0 || d->kw <= 0 || d->div_shift < 0 || d->div_shift > 2) return RN_EINVAL;
    if (d->add_mode < 0 || d->add_mode > 2 || d->act < 0 || d->act > 2) return RN_EINVAL;
    if (d->mask_mode < 0 || (d->mask_mode & ~(3 | RN_MASK_BITS)) || (d->mask_mode & 3) == 3 || d->mask_mode == RN_MASK_BITS) return RN_EINVAL;
    if (((d->mask_mode & RN_MASK_BITS) || d->sign_out != nullptr) && ((d->Cout & 31) || (d->y_batch_stride & 31))) return RN_EINVAL;
    if (d->os < 1 || d->oo_h < 0 || d->oo_w < 0) return RN_EINVAL;
    if ((d->Ho - 1) * d->os + d->oo_h >= d->Hy || (d->Wo - 1) * d->os + d->oo_w >= d->Wy) return RN_EINVAL;
    if (d->os != 1 && d->add_mode == 2) return RN_EINVAL;
    if (d->in_relu || d->add2_mode != 0 || d->w_batch_stride != 0) return RN_EINVAL;   // not in the bf16 form (yet)
    return RN_OK;
}

static inline int check_ptrs_bf16(const rn_conv_desc *d, const void *x, const void *y, const void *add, const void *mask, int y_is_f32) {
    if ((d->add_mode != 0) != (add != nullptr) || (d->mask_mode != 0) != (mask != nullptr)) return RN_EINVAL;
    if (!y_is_f32 && (d->Cout & 7)) return RN_EINVAL;                    // bf16 result: 8 channels = 16 bytes per lane
    const uintptr_t am = y_is_f32 ? 7 : 15;                               // addend / mask: 8 bytes beside an fp32 result, else 16
    const uintptr_t mm = (d->mask_mode & RN_MASK_BITS) ? 3 : am;           // sign-bit words
    if (d->sign_out != nullptr && (y_is_f32 || ((uintptr_t)d->sign_out & 3))) return RN_EINVAL;   // sign bits: bf16 results only
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)add & am) || ((uintptr_t)mask & mm)) return RN_EINVAL;
    if (!y_is_f32 && ((d->y_batch_stride & 7) || (d->add_batch_stride & 7))) return RN_EINVAL;
    return RN_OK;
}

// (Rounds 2-4 also carried a 256 x 256 tile of sixteen waves, one 1024-thread workgroup per CU, behind RN_BF16_BIG_TILE: +25 % on long-K 3x3
// layers alone, -5 % on the whole step (171.1 against 180.0 images/s), and superseded by the eight-wave phased kernel of conv_bf16_p8.hip
// in round 4.  Its instances left the library in round 5; the 256 x 256 tile rows below are the phased kernel's.)
static inline bool bf16_big_tile(int64_t, int, int, int) { return false; }
static inline bool bf16_tile_is_big(const rn_conv_group *, int) { return false; }
// 256 x 128 tile for a group: dense bf16 results, Cout a multiple of 128, a long K loop and enough tiles (see the single launcher)
static inline bool bf16_group_is_tall(const rn_conv_group *g, int y_is_f32) {
    static const int tall_env = [] { const char *e = getenv("RN_BF16_TALL_TILE"); return e ? atoi(e) : -1; }();
    if (y_is_f32 || tall_env == 0 || bf16_tile_is_big(g, y_is_f32) || (g->d[0].Cout & 127) != 0) return false;
    int64_t t = 0;
    for (int i = 0; i < g->n; ++i) {
        if (!bf16_desc_is_dense(&g->d[i])) return false;
        t += (((int64_t)g->d[i].N * g->d[i].Ho * g->d[i].Wo + 255) / 256) * ((g->d[i].Cout + 127) / 128);
    }
    return tall_env == 1 || (g->d[0].kh * g->d[0].kw * g->d[0].Cin >= 1024 && t >= 512);
}
// Tile shape rn_conv_igemm_bf16_grouped will use for this group: the caller builds tile_end with it
// (tile_end[i] = running sum of ceil(N*Ho*Wo / rows) * ceil(Cout / cols)).  Returns rows * 1000 + cols.
extern "C" int rn_conv_igemm_bf16_tile_rows(const rn_conv_group *g, int y_is_f32) {
    if (g->n < 1 || g->n > RN_MAX_GROUP) return 0;
    if (bf16_group_is_p8(g, y_is_f32)) return 256 * 1000 + 256;
    if (bf16_tile_is_big(g, y_is_f32)) return 256 * 1000 + 256;
    if (bf16_group_is_tall(g, y_is_f32)) return 256 * 1000 + 128;
    return 128 * 1000 + 128;
}

// The tile a SINGLE launch (rn_conv_igemm_bf16) takes: rows * 1000 + cols, + 1 000 000 for the eight-wave phased kernel (profiling / tests).
extern "C" int rn_conv_igemm_bf16_tile(const rn_conv_desc *d, int y_is_f32) {
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    if (bf16_p8_pick(d, y_is_f32, ((M + 255) / 256) * ((d->Cout + 255) / 256))) return 1000000 + 256 * 1000 + 256;
    if (bf16_big_tile(((M + 255) / 256) * ((d->Cout + 255) / 256), d->Cout, d->kh * d->kw * d->Cin, y_is_f32)) return 256 * 1000 + 256;
    return 128 * 1000 + 128;                                  // (or 256 x 128 for long dense launches: rn_conv_igemm_bf16)
}

extern "C" int rn_conv_igemm_bf16_grouped(const rn_conv_group *g, const void *w_packed, int y_is_f32, const float *scale,
                                          const float *shift, void *stream) {
    if (g->n < 1 || g->n > RN_MAX_GROUP || ((uintptr_t)w_packed & 15)) return RN_EINVAL;
    const rn_conv_desc &d0 = g->d[0];
    const bool p8 = bf16_group_is_p8(g, y_is_f32);
    const bool big = p8 || bf16_tile_is_big(g, y_is_f32);    // the caller's tile_end must follow rn_conv_igemm_bf16_tile_rows()
    const bool tall = !p8 && bf16_group_is_tall(g, y_is_f32);
    const int TR = big ? 256 : 128, TRM = (big || tall) ? 256 : 128;
    int prev = 0;
    for (int i = 0; i < g->n; ++i) {
        const rn_conv_desc &d = g->d[i];
        int rc = check_desc_bf16(&d);
        if (rc) return rc;
        rc = check_ptrs_bf16(&d, g->x[i], g->y[i], g->add[i], g->mask[i], y_is_f32);
        if (rc) return rc;
        if (d.Cin != d0.Cin || d.Cout != d0.Cout || d.kh != d0.kh || d.kw != d0.kw || d.act != d0.act) return RN_EINVAL;
        const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
        const int64_t tiles = ((M + TRM - 1) / TRM) * ((d.Cout + TR - 1) / TR);
        if (g->tile_end[i] - prev != tiles) return RN_EINVAL;
        prev = g->tile_end[i];
    }
    const __bf16 *wb = reinterpret_cast<const __bf16 *>(w_packed);
    bool dense = true;
    for (int i = 0; i < g->n; ++i) dense = dense && bf16_desc_is_dense(&g->d[i]);
    if (p8) return rn_bf16_p8_launch_grouped(g, prev, w_packed, scale, shift, (hipStream_t)stream);
    {
        const dim3 grid((unsigned)prev), block(256);
        if (y_is_f32) hipLaunchKernelGGL((conv_igemm_bf16_grouped_kernel<true, 2, 2, false>), grid, block, 0, (hipStream_t)stream, *g, wb, scale, shift);
        else if (tall) hipLaunchKernelGGL((conv_igemm_bf16_grouped_kernel<false, 2, 2, true, 4>), grid, block, 0, (hipStream_t)stream, *g, wb, scale, shift);
        else if (dense) hipLaunchKernelGGL((conv_igemm_bf16_grouped_kernel<false, 2, 2, true>), grid, block, 0, (hipStream_t)stream, *g, wb, scale, shift);
        else hipLaunchKernelGGL((conv_igemm_bf16_grouped_kernel<false, 2, 2, false>), grid, block, 0, (hipStream_t)stream, *g, wb, scale, shift);
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_conv_igemm_bf16(const rn_conv_desc *d, const void *x, const void *w_packed, void *y, int y_is_f32,
                                  const float *scale, const float *shift, const void *add, const void *mask, void *stream) {
    const int rc = check_desc_bf16(d);
    if (rc) return rc;
    const int rp = check_ptrs_bf16(d, x, y, add, mask, y_is_f32);
    if (rp || ((uintptr_t)w_packed & 15)) return RN_EINVAL;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    if (bf16_p8_pick(d, y_is_f32, ((M + 255) / 256) * ((d->Cout + 255) / 256)))
        return rn_bf16_p8_launch(d, x, w_packed, y, scale, shift, add, mask, (hipStream_t)stream);
    const bool big = bf16_big_tile(((M + 255) / 256) * ((d->Cout + 255) / 256), d->Cout, d->kh * d->kw * d->Cin, y_is_f32);
    // 256 x 128 tile (4 waves of 128 x 64, two workgroups per CU): 85 FLOP per staged byte instead of 64, for dense bf16
    // results with a long K loop and enough tiles (RN_BF16_TALL_TILE=0 turns it off, =1 forces it where it is legal)
    static const int tall_env = [] { const char *e = getenv("RN_BF16_TALL_TILE"); return e ? atoi(e) : -1; }();
    const int64_t tall_tiles = ((M + 255) / 256) * ((d->Cout + 127) / 128);
    const bool tall_ok = !big && !y_is_f32 && bf16_desc_is_dense(d) && (d->Cout & 127) == 0;
    const bool tall = tall_ok && (tall_env == 1 || (tall_env != 0 && d->kh * d->kw * d->Cin >= 1024 && tall_tiles >= 512));
    // 256 x 64 tile (4 x 1 waves of 64 x 64) for dense bf16 results with at most 64 output channels: the 128 x 128 tile computes half of
    // its columns for nothing there (3x3 64 -> 64, 1x1 256 -> 64 and their data gradients).  RN_BF16_NARROW_TILE=0 turns it off (A/B).
    static const int narrow_env = [] { const char *e = getenv("RN_BF16_NARROW_TILE"); return e ? atoi(e) : 1; }();
    const bool narrow = narrow_env != 0 && !big && !tall && !y_is_f32 && d->Cout <= 64 && bf16_desc_is_dense(d);
    const int TR = big ? 256 : (narrow ? 64 : 128), TRM = (big || tall || narrow) ? 256 : 128;
    const int64_t tiles = ((M + TRM - 1) / TRM) * ((d->Cout + TR - 1) / TR);
    if (tiles > 0x7fffffff) return RN_EINVAL;
    const dim3 grid((unsigned)tiles), block(big ? 1024 : 256);
    const __bf16 *xb = reinterpret_cast<const __bf16 *>(x), *wb = reinterpret_cast<const __bf16 *>(w_packed);
    const __bf16 *ab = reinterpret_cast<const __bf16 *>(add), *mb = reinterpret_cast<const __bf16 *>(mask);
    if (tall) hipLaunchKernelGGL((conv_igemm_bf16_kernel<false, 2, 2, true, 4>), grid, block, 0, (hipStream_t)stream, *d, xb, wb, y, scale, shift, ab, mb);
    else if (narrow) hipLaunchKernelGGL((conv_igemm_bf16_kernel<false, 4, 1, true>), grid, block, 0, (hipStream_t)stream, *d, xb, wb, y, scale, shift, ab, mb);
    else if (y_is_f32) hipLaunchKernelGGL((conv_igemm_bf16_kernel<true, 2, 2, false>), grid, block, 0, (hipStream_t)stream, *d, xb, wb, y, scale, shift, ab, mb);
    else if (bf16_desc_is_dense(d)) hipLaunchKernelGGL((conv_igemm_bf16_kernel<false, 2, 2, true>), grid, block, 0, (hipStream_t)stream, *d, xb, wb, y, scale, shift, ab, mb);
    else hipLaunchKernelGGL((conv_igemm_bf16_kernel<false, 2, 2, false>), grid, block, 0, (hipStream_t)stream, *d, xb, wb, y, scale, shift, ab, mb);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ---------------------------------------------------------------------------------------------- weight gradient
// dw[co][kk] += sum over pixels of dy[pixel][co] * x[pixel shifted by tap(kk)][ci(kk)], kk = packed-row index
// (tap-major, channel-minor).  Tile 128 (co) x 128 (kk), 32 pixels per K-step, K split over the grid, fp32 atomics.
struct WgradBf16Args {
    const __bf16 *dy, *x;
    float *dw;
    float *colsum;               // [Cout] += sum over pixels of dy, or NULL
    int ldy, N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad;
    int Kflat, Kpad;             // kh*kw*Cin and its round-up to 32
    int tiles_n, tiles, splits;
    int xcd_map;                 // 1: whole K slices per XCD (the kernel)
    int64_t pixels, per_split;   // K extent and K per slice (multiple of 32)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
#define WG_WK 32                 // pixels per K-step
#define WG_ROWB 256              // bytes per staged row: 128 bf16

// LDS image of a staged [32 pixels][128 columns] bf16 tile: 256-byte rows whose 16 chunks of 16 bytes are XOR-permuted
// by wg_fx(row), the image that serves gfx950's transposing read without bank conflicts (cdna_hip_programming.md T10 (b);
// operand addressing verified by tools/probes/tr_probe.hip).  The direct-to-LDS load fills rows linearly, so the
// permutation is applied on the load's SOURCE: the lane filling position cp of row r fetches logical chunk cp ^ wg_fx(r).
__device__ __forceinline__ int wg_fx(int row) { return WgradBf16Geom::fx(row); }

// One 32x32x16 operand from such an image: lane l <- 8 consecutive rows (pixels) 16*kh + 8*(l>>5) + 0..7 of column
// col0 + (l & 31), by two ds_read_b64_tr_b16 (each: a 4-row x 16-column block per 16 lanes, returned column-major).
__device__ __forceinline__ bf16x8 wg_operand(const char *img, unsigned a_rd0, unsigned a_rd1, int kh) {
    typedef __attribute__((address_space(3))) s16x4 *lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(__attribute__((address_space(3))) char *)(img + a_rd0 + kh * 16 * WG_ROWB));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(__attribute__((address_space(3))) char *)(img + a_rd1 + kh * 16 * WG_ROWB));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// bid: the workgroup's id within ITS problem's range of the grid (a grouped launch holds several problems' ranges one after the other)
__device__ __forceinline__ void conv_wgrad_bf16_body(const WgradBf16Args &p, const int bid) {
    constexpr int BM = 128, BN = 128, WK = WG_WK;
    constexpr int TB = 256 / WK;                             // K-steps per pixel-table batch: one entry per thread
    constexpr int IA = WK / 4 / 4, IB = WK / 4 / 4;          // DMA instructions per wave per K-step: 4 rows each, 4 waves
    __shared__ __attribute__((aligned(16))) char lds[2][WK * (BM + BN) * 2];   // per buffer: dY [32][128], then X [32][128]
    __shared__ int4 pixtab[2][TB * WK];                      // (x byte offset of input pixel (ih0, iw0), ih0, iw0, -)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup id -> (tile, K slice).  Ids are dealt round-robin to the 8 XCDs; all tiles of a slice stream the same dY / X
    // slabs, so a slice is kept on ONE XCD (slice s on XCD s % 8, its tiles consecutive there): the slabs come through that
    // L2 once instead of through all eight.  Without it the PMC counters showed an L2 hit rate of 39 % and 5.5x the
    // algorithmic bytes fetched (profiles/r02_pmc_conv_bf16.txt); conv_wgrad.hip does the same for its many-tile shapes.
    int slice, tile;
    if (p.xcd_map) {
        const int xcd = bid & 7, wi = bid >> 3;                // a problem's range starts at a multiple of 8: bid & 7 is the XCD
        slice = (wi / p.tiles) * 8 + xcd;
        tile = wi % p.tiles;
    } else {
        slice = bid / p.tiles;
        tile = bid % p.tiles;
    }
    if (slice >= p.splits) return;                           // padding of the last group of 8 slices
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int64_t kbeg = (int64_t)slice * p.per_split;
    const int64_t kend = (kbeg + p.per_split < p.pixels) ? kbeg + p.per_split : p.pixels;
    const int nks = (int)((kend - kbeg + WK - 1) / WK);
    if (nks <= 0) return;
    const int HoWo = p.Ho * p.Wo;

    // buffer descriptors: dY = this K slice only (pixels past kend are out of range by themselves); X from the first
    // image the slice touches (the host keeps a slice's span of images below 2 GiB)
    const int n_first = (int)(kbeg / HoWo);
    const int rel0 = (int)(kbeg - (int64_t)n_first * HoWo);
    const int64_t img = (int64_t)p.Hi * p.Wi * p.Cin;
    int64_t xbytes = ((int64_t)p.N - n_first) * img * 2;
    if (xbytes > 0x7FFFFFFF) xbytes = 0x7FFFFFFF;
    const v4i32 rs_a = make_rsrc(p.dy + kbeg * p.ldy, (unsigned)((kend - kbeg) * p.ldy * 2));
    const v4i32 rs_b = make_rsrc(p.x + (int64_t)n_first * img, (unsigned)xbytes);

    // staging geometry: instruction j of wave w fills pixel rows (w*I + j)*4 + lane/16; the lane fills chunk position cp
    const int cp = lane & 15, rq = lane >> 4;
    unsigned a_voff[IA];
    int b_fr[IB], b_fs[IB], b_tapoff[IB];
#pragma unroll
    for (int j = 0; j < IA; ++j) {
        const int row = WgradBf16Geom::dma_row(wave, j) + rq;
        const int col = m0 + 8 * (cp ^ wg_fx(row));
        a_voff[j] = col < p.ldy ? (unsigned)((row * p.ldy + col) * 2) : 0x80000000u;
    }
#pragma unroll
    for (int j = 0; j < IB; ++j) {
        const int row = WgradBf16Geom::dma_row(wave, j) + rq;
        const int jcol = n0 + 8 * (cp ^ wg_fx(row));         // the chunk fixes (tap, first input channel)
        const int tap = jcol / p.Cin;
        const int ci0 = jcol - tap * p.Cin;
        const int fr = tap / p.kw;
        b_fs[j] = tap - fr * p.kw;
        b_fr[j] = jcol < p.Kflat ? fr : (1 << 24);           // a column past the matrix fails every row test
        b_tapoff[j] = ((fr * p.Wi + b_fs[j]) * p.Cin + ci0) * 2;
    }

    auto fill_batch = [&](int j) {
        const int rel = rel0 + j * (TB * WK) + tid;
        int4 e = make_int4(0, -(1 << 28), 0, 0);             // past the slice: fails the row test
        if (kbeg + (int64_t)j * (TB * WK) + tid < kend) {
            const unsigned n = (unsigned)rel / (unsigned)HoWo;
            const unsigned rem = (unsigned)rel - n * (unsigned)HoWo;
            const unsigned oh = rem / (unsigned)p.Wo, ow = rem - oh * (unsigned)p.Wo;
            const int ih0 = (int)oh * p.stride - p.pad, iw0 = (int)ow * p.stride - p.pad;
            e = make_int4((((int)n * p.Hi + ih0) * p.Wi + iw0) * p.Cin * 2, ih0, iw0, 0);
        }
        pixtab[j & 1][tid] = e;
    };
    unsigned b_voff[IB];
    auto make_offsets = [&](int ks) {
        const int4 *tab = pixtab[(ks / TB) & 1];
#pragma unroll
        for (int j = 0; j < IB; ++j) {
            const int4 e = tab[WgradBf16Geom::tab_index(ks, wave, lane, j)];
            const bool ok = ((unsigned)(e.y + b_fr[j]) < (unsigned)p.Hi) & ((unsigned)(e.z + b_fs[j]) < (unsigned)p.Wi);
            b_voff[j] = ok ? (unsigned)(e.x + b_tapoff[j]) : 0xFFFFFFFFu;
        }
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds0 = lds_addr(&lds[0][0]);
    constexpr unsigned BUFB = WK * (BM + BN) * 2;
    auto dma_step = [&](int ks, int buf) {
        const unsigned A = lds0 + (unsigned)(buf * BUFB + wave_u * IA * 4 * WG_ROWB);
        const unsigned B = lds0 + (unsigned)(buf * BUFB + WK * WG_ROWB + wave_u * IB * 4 * WG_ROWB);
        const unsigned so = (unsigned)ks * (unsigned)(WK * 2) * (unsigned)p.ldy;     // scalar: the K-step advance of dY
#pragma unroll
        for (int j = 0; j < IA; ++j) dma16(rs_a, A + j * (4 * WG_ROWB), a_voff[j], so);   // a_voff[j] holds instruction j's rows
#pragma unroll
        for (int j = 0; j < IB; ++j) dma16(rs_b, B + j * (4 * WG_ROWB), b_voff[j], 0u);
    };

    // fragment read addresses (bytes within a buffer): block row r0 + q, chunk c0 + (pp >> 1), half pp & 1 (conv_wgrad_geom.h)
    unsigned fa[2][2], fb[2][2];                             // [32-column sub-tile][read 0/1]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            fa[t][rd] = (unsigned)WgradBf16Geom::tr_addr(wm, t, rd, 0, lane);
            fb[t][rd] = (unsigned)(WgradBf16Geom::IMG + WgradBf16Geom::tr_addr(wn, t, rd, 0, lane));
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e_ = 0; e_ < 16; ++e_) acc[i][j][e_] = 0.f;
    float cs[2] = {0.f, 0.f};
    const bool do_cs = p.colsum != nullptr && (tile % p.tiles_n) == 0 && wn == 0;

    fill_batch(0);
    __syncthreads();
    make_offsets(0);
    dma_step(0, 0);
    rn_wait_dma();
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) {
            // the table batch of step ks+1 was published by an earlier barrier: batch b is filled during step
            // b*TB - 4 (below) or before the loop (batch 0)
            make_offsets(ks + 1);
            dma_step(ks + 1, buf ^ 1);
        }
        const char *S = &lds[buf][0];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const bf16x8 a0 = wg_operand(S, fa[0][0], fa[0][1], kh), a1 = wg_operand(S, fa[1][0], fa[1][1], kh);
            const bf16x8 b0 = wg_operand(S, fb[0][0], fb[0][1], kh), b1 = wg_operand(S, fb[1][0], fb[1][1], kh);
            if (do_cs) {
#pragma unroll
                for (int e_ = 0; e_ < 8; ++e_) { cs[0] += (float)a0[e_]; cs[1] += (float)a1[e_]; }
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        // next table batch: needed from step (b+1)*TB - 1 on (make_offsets(ks + 1)); its slot held batch b-1, last read
        // for step b*TB - 1, i.e. during iteration b*TB - 2
        if ((ks % TB) == TB - 4 && (ks / TB + 1) * TB < nks) fill_batch(ks / TB + 1);
        rn_wait_dma();
        __syncthreads();
    }
    if (do_cs) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            cs[t] += __shfl_xor(cs[t], 32);                   // the two 8-pixel groups of the same channel
            const int c = m0 + wm * 64 + t * 32 + (lane & 31);
            if (lane < 32 && c < p.Cout) atomicAdd(p.colsum + c, cs[t]);
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
                if (row < p.Cout && col < p.Kflat) RN_WG_ATOMIC(p.dw + (int64_t)row * p.Kpad + col, acc[tm][tn][e]);
            }
        }
}

__global__ __launch_bounds__(256, 3) void conv_wgrad_bf16_kernel(const WgradBf16Args p) { conv_wgrad_bf16_body(p, (int)blockIdx.x); }

// Grouped launch: the pyramid levels of a head layer (D/model.py:110-205: one weight tensor convolves all five levels, so its gradient
// is the SUM over the levels) as ONE grid accumulating into one dw / colsum.  Alone, the small levels are launches of a few dozen
// microseconds at 60-270 TFLOP/s (8 x 9 x 15 ... 8 x 34 x 60 pixels); here they ride along with the big ones.
struct WgradBf16Group {
    int n;
    int block_end[RN_MAX_GROUP];         // running sum of the problems' workgroup counts (each a multiple of 8)
    WgradBf16Args p[RN_MAX_GROUP];
};
__global__ __launch_bounds__(256, 3) void conv_wgrad_bf16_grouped_kernel(const WgradBf16Group g) {
    int i = 0;
#pragma unroll
    for (int j = 0; j < RN_MAX_GROUP - 1; ++j) i += (j + 1 < g.n && (int)blockIdx.x >= g.block_end[j]) ? 1 : 0;
    i = __builtin_amdgcn_readfirstlane(i);
    conv_wgrad_bf16_body(g.p[i], (int)blockIdx.x - (i > 0 ? g.block_end[i - 1] : 0));
}

// Geometry, K slices and grid size of one weight-gradient problem; target_wgs: the workgroups it should bring to the grid.
// Returns the number of workgroups (> 0) or -RN_EINVAL.
static int64_t wgrad_bf16_fill(WgradBf16Args &a, const void *dy, int ldy, const void *x, float *dw, float *colsum, int N, int Hi, int Wi,
                               int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int target_wgs, bool pad8) {
    a.dy = reinterpret_cast<const __bf16 *>(dy); a.x = reinterpret_cast<const __bf16 *>(x); a.dw = dw; a.colsum = colsum; a.ldy = ldy;
    a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
    a.Kflat = kh * kw * Cin;
    a.Kpad = (a.Kflat + 31) / 32 * 32;
    a.pixels = (int64_t)N * Ho * Wo;
    const int tiles_m = (Cout + 127) / 128;
    a.tiles_n = (a.Kflat + 127) / 128;
    a.tiles = tiles_m * a.tiles_n;
    int64_t splits = (target_wgs + a.tiles - 1) / a.tiles;
    const int64_t max_splits = (a.pixels + 16 * WG_WK - 1) / (16 * WG_WK);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const int64_t HoWo = (int64_t)Ho * Wo, img_bytes = (int64_t)Hi * Wi * Cin * 2;
    for (;;) {
        a.per_split = ((a.pixels + splits - 1) / splits + WG_WK - 1) / WG_WK * WG_WK;
        const int64_t span_imgs = (a.per_split + HoWo - 2) / HoWo + 1;
        if ((a.per_split + WG_WK) * ldy * 2 <= 0x7FFFFFFF && span_imgs * img_bytes <= 0x7FFFFFFF) break;
        if (a.per_split <= WG_WK || splits >= 65535) return -RN_EINVAL;
        splits = splits * 2 > 65535 ? 65535 : splits * 2;
    }
    splits = (a.pixels + a.per_split - 1) / a.per_split;
    a.splits = (int)splits;
    static const int xcd_env = [] { const char *e = getenv("RN_WGRAD_BF16_XCD"); return e ? atoi(e) : 1; }();
    a.xcd_map = xcd_env != 0 && a.tiles >= 4 && splits >= 8;
    int64_t blocks = a.tiles * (a.xcd_map ? (splits + 7) / 8 * 8 : splits);
    if (pad8 && !a.xcd_map) blocks = (blocks + 7) / 8 * 8;      // grouped: every problem's range starts at a multiple of 8 (surplus ids: slice >= splits)
    return blocks;
}

extern "C" int rn_conv_wgrad_bf16(const void *dy, int ldy, const void *x, float *dw, float *colsum, int N, int Hi, int Wi,
                                  int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, void *stream) {
    if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || Cin < 8 || (Cin & 7) || (ldy & 7) || ldy < Cout)
        return RN_EINVAL;
    if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15)) return RN_EINVAL;
    // (A 256 x 256 eight-wave form of this reduction was built in round 4, correct and slower -- 0.51 against 0.40 ms on the head tower
    // layer, profiles/r04_wgrad_p8_knockouts.txt -- and now lives in tools/probes/quarantine_r05/.)
    static const int target_wgs = [] { const char *e = getenv("RN_WGRAD_BF16_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 768; }();
    // K slices for ~768 workgroups = ONE resident round at three per CU.  Every slice ends in tile-sized fp32 atomics
    // (Cout x Kflat x slices of them per launch, ~1.3 TB/s chip-wide), and with the MFMAs eight times shorter than in the
    // fp32 kernel that tail weighs more: measured per training step (all weight gradients) 512: 11.6 ms, 768: 11.1,
    // 1024: 13.5, 1536: 13.2, 2048: 14.6, 3072: 15.2 (the fp32 kernel's optimum is 2048).  RN_WGRAD_BF16_WGS overrides.
    WgradBf16Args a;
    const int64_t blocks = wgrad_bf16_fill(a, dy, ldy, x, dw, colsum, N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad, target_wgs, false);
    if (blocks <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(conv_wgrad_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// The weight gradients of n <= RN_MAX_GROUP problems that share ONE weight tensor (the pyramid levels of a head layer) as one launch:
// dw / colsum accumulate the sum over the problems.  Per problem i: dy[i] [N, H[i], W[i], ldy] and x[i] [N, H[i], W[i], Cin] (stride-1
// same-size geometry is NOT required: Ho / Wo follow from H, W, k, stride, pad as in rn_conv_wgrad_bf16).
extern "C" int rn_conv_wgrad_bf16_grouped(int n, const void *const *dy, int ldy, const void *const *x, float *dw, float *colsum, int N,
                                          const int *Hi, const int *Wi, int Cin, int Cout, int kh, int kw, int stride, int pad, void *stream) {
    if (n < 1 || n > RN_MAX_GROUP || N <= 0 || Cout <= 0 || Cin < 8 || (Cin & 7) || (ldy & 7) || ldy < Cout) return RN_EINVAL;
    static const int target_wgs = [] { const char *e = getenv("RN_WGRAD_BF16_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 768; }();
    int Ho[RN_MAX_GROUP], Wo[RN_MAX_GROUP];
    int64_t total_pixels = 0;
    for (int i = 0; i < n; ++i) {
        if (Hi[i] <= 0 || Wi[i] <= 0 || ((uintptr_t)dy[i] & 15) || ((uintptr_t)x[i] & 15)) return RN_EINVAL;
        Ho[i] = (Hi[i] + 2 * pad - kh) / stride + 1;
        Wo[i] = (Wi[i] + 2 * pad - kw) / stride + 1;
        if (Ho[i] <= 0 || Wo[i] <= 0) return RN_EINVAL;
        total_pixels += (int64_t)N * Ho[i] * Wo[i];
    }
    WgradBf16Group g;
    g.n = n;
    int64_t end = 0;
    for (int i = 0; i < n; ++i) {
        // the problem's share of one resident round, by its pixels (at least one slice per tile: wgrad_bf16_fill)
        const int share = (int)((int64_t)target_wgs * ((int64_t)N * Ho[i] * Wo[i]) / total_pixels);
        const int64_t blocks = wgrad_bf16_fill(g.p[i], dy[i], ldy, x[i], dw, colsum, N, Hi[i], Wi[i], Cin, Ho[i], Wo[i], Cout, kh, kw, stride, pad,
                                               share, true);
        if (blocks <= 0) return RN_EINVAL;
        end += blocks;
        if (end > 0x7fffffff) return RN_EINVAL;
        g.block_end[i] = (int)end;
    }
    for (int i = n; i < RN_MAX_GROUP; ++i) g.block_end[i] = (int)end;
    hipLaunchKernelGGL(conv_wgrad_bf16_grouped_kernel, dim3((unsigned)end), dim3(256), 0, (hipStream_t)stream, g);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
