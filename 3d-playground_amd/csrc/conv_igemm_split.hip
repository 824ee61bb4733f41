// The implicit-GEMM convolution with split-operand products (mfma_split.h): the tile of conv_igemm_tile.h with its eight
// v_mfma_f32_32x32x2_f32 per 32x32x16 block replaced by six v_mfma_f32_32x32x16_bf16 on (h, m, l) splits of the fp32
// fragments.  Same descriptors, same packed fp32 weights, same staging and epilogue as conv_igemm.hip, whose launchers
// choose between the two (rn_get_fp32_mfma()).  Kept in its own translation unit: every instance of the force-inlined tile
// costs compile time and memory.
//
// K-step: 16, like the fp32 form.  32 (template parameter BK; two MFMA steps per barrier, 64-80 KB of LDS, two workgroups per CU)
// was measured on the training step: 87.5 against 89.3 images/s on the same GPU, and no layer shape gained in isolation.
//
// One tile per workgroup, also for the plain GEMMs of the Winograd layers (thousands of 16-step tiles per launch): letting a
// workgroup run 2 / 4 consecutive tiles, so that a tile's stores drain under the next one's set-up, measured 87.6 / 86.3 against
// 88.1 images/s.
//
// Register budget: the fragments (32 floats), their splits (48 registers) and the 64 accumulators do not fit the 128
// registers four workgroups per CU leave, so these kernels run three per CU (168 registers, 34-41 KB of LDS each); forced to
// four (13-18 spilled registers) the implicit-GEMM family takes 46.3 instead of 42.6 ms per training step.
#include "conv_igemm_tile.h"

template <int WM, int WN, bool GENERAL, bool RELU, bool RAW, int SPLIT, int BK = 16, int TERMS = 3>
__global__ __launch_bounds__(256, BK == 32 ? 2 : 3) void conv_igemm_split_kernel(const rn_conv_desc d, const float *__restrict__ x,
                                                                  const float *__restrict__ w, float *__restrict__ y,
                                                                  const float *__restrict__ scale, const float *__restrict__ shift,
                                                                  const float *__restrict__ add, const float *__restrict__ mask,
                                                                  const float *__restrict__ add2) {
    conv_igemm_tile<WM, WN, GENERAL, BK, RELU, RAW, SPLIT, TERMS>(d, x, w, y, scale, shift, add, mask, add2, xcd_remap(blockIdx.x, gridDim.x));
}

template <int WM, int WN, int SPLIT, int BK = 16, int TERMS = 3>
__global__ __launch_bounds__(256, BK == 32 ? 2 : 3) void conv_igemm_split_grouped_kernel(const rn_conv_group g, const float *__restrict__ w,
                                                                          const float *__restrict__ scale,
                                                                          const float *__restrict__ shift) {
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
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
    conv_igemm_tile<WM, WN, true, BK, false, false, SPLIT, TERMS>(d, x, w, y, scale, shift, add, mask, nullptr, tile - first);
}

// SPLIT 3 (conv_igemm_tile.h: the activation operand split once per workgroup into bf16 planes in LDS) for the 128 x 128 tile
// when the weights come pre-split and a K-step lies inside one filter tap; RN_SPLIT_A_ONCE=0 keeps the SPLIT 2 kernels (A/B).
static bool split_a_once(const rn_conv_desc *d) {
    static const int on = [] { const char *e = getenv("RN_SPLIT_A_ONCE"); return e ? atoi(e) : 1; }();
    return on && d->w_format == 1 && (d->Cin % 16) == 0 && d->div_shift == 0 && d->kh * d->kw <= 24;
}

// conv_igemm_mf16.hip: 128 x 128 tiles on v_mfma_f32_16x16x32_bf16 (RN_FP32_SPLIT) / v_mfma_f32_16x16x32_f16 (RN_FP32_SPLIT3)
bool rn_igemm_mf16_launch(int variant, const rn_conv_desc *d, const float *x, const float *w, float *y, const float *scale,
                          const float *shift, const float *add, const float *mask, const float *add2, hipStream_t s, int *rc);
bool rn_igemm_mf16_grouped_launch(const rn_conv_group *g, const float *w, const float *scale, const float *shift, hipStream_t s, int *rc);

static int dbg_dyn_lds(const void *fn) {                  // occupancy experiment: RN_DBG_DYN_LDS bytes of unused dynamic LDS per workgroup
    static const int v = getenv("RN_DBG_DYN_LDS") ? atoi(getenv("RN_DBG_DYN_LDS")) : 0;
    if (v > 16384) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, v);
    return v;
}

// variant: the instance conv_igemm.hip's launcher chose -- 0 raw GEMM, 1 input ReLU, 2 / 3 narrow dense / general,
// 4 / 5 wide dense / general.  d->w_format 1: w is the pre-split form (rn_split_weights).
int rn_igemm_split_launch(int variant, unsigned tiles, const rn_conv_desc *d, const float *x, const float *w, float *y,
                          const float *scale, const float *shift, const float *add, const float *mask, const float *add2,
                          hipStream_t s) {
    if (d->w_format == 2) {
        // pre-split weights, products from the first bf16 terms only (the fp32 stem of the bf16 / fp8 engines): the narrow instances
        const dim3 grid1(tiles), block1(256);
        if (variant == 2) hipLaunchKernelGGL((conv_igemm_split_kernel<4, 1, false, false, false, 2, 16, 1>), grid1, block1, 0, s, *d, x, w, y, scale, shift, add, mask, add2);
        else if (variant == 3) hipLaunchKernelGGL((conv_igemm_split_kernel<4, 1, true, false, false, 2, 16, 1>), grid1, block1, 0, s, *d, x, w, y, scale, shift, add, mask, add2);
        else return RN_EINVAL;
        RN_LAUNCH_CHECK();
        return RN_OK;
    }
    {
        int rc = RN_OK;
        if (rn_igemm_mf16_launch(variant, d, x, w, y, scale, shift, add, mask, add2, s, &rc)) return rc;
    }
    if (d->w_format == 3) {
        // RN_FP32_SPLIT3 for what the 16x16x32 kernel does not take (<= 64 output channels, the 4-channel stem, channel counts that are
        // no multiple of 32, the input-ReLU form): the tile of conv_igemm_tile.h with two-term fp16 products (TERMS 2)
        if (d->x_amax == nullptr || d->w_unscale == nullptr) return RN_EINVAL;
        const dim3 grid2(tiles), block2(256);
#define RN_HALF_LAUNCH(WM, WN, G, R, RAW) \
        hipLaunchKernelGGL((conv_igemm_split_kernel<WM, WN, G, R, RAW, 2, 16, 2>), grid2, block2, 0, s, *d, x, w, y, scale, shift, add, mask, add2)
        switch (variant) {
            case 0: RN_HALF_LAUNCH(2, 2, false, false, true); break;
            case 1: RN_HALF_LAUNCH(2, 2, true, true, false); break;
            case 2: RN_HALF_LAUNCH(4, 1, false, false, false); break;
            case 3: RN_HALF_LAUNCH(4, 1, true, false, false); break;
            case 4: RN_HALF_LAUNCH(2, 2, false, false, false); break;
            case 5: RN_HALF_LAUNCH(2, 2, true, false, false); break;
            default: return RN_EINVAL;
        }
#undef RN_HALF_LAUNCH
        RN_LAUNCH_CHECK();
        return RN_OK;
    }
    const dim3 grid(tiles), block(256);
#define RN_SPLIT_LAUNCH(WM, WN, G, R, RAW)                                                                                              \
    do {                                                                                                                                \
        if (WM == 2 && split_a_once(d)) hipLaunchKernelGGL((conv_igemm_split_kernel<WM, WN, G, R, RAW, (WM == 2 ? 3 : 2)>), grid, block, dbg_dyn_lds((const void *)conv_igemm_split_kernel<WM, WN, G, R, RAW, (WM == 2 ? 3 : 2)>), s, *d, x, w, y, scale, shift, add, mask, add2); \
        else if (d->w_format == 1) hipLaunchKernelGGL((conv_igemm_split_kernel<WM, WN, G, R, RAW, 2>), grid, block, 0, s, *d, x, w, y, scale, shift, add, mask, add2); \
        else hipLaunchKernelGGL((conv_igemm_split_kernel<WM, WN, G, R, RAW, 1>), grid, block, 0, s, *d, x, w, y, scale, shift, add, mask, add2);                 \
    } while (0)
    switch (variant) {
        case 0: RN_SPLIT_LAUNCH(2, 2, false, false, true); break;
        case 1: RN_SPLIT_LAUNCH(2, 2, true, true, false); break;
        case 2: RN_SPLIT_LAUNCH(4, 1, false, false, false); break;
        case 3: RN_SPLIT_LAUNCH(4, 1, true, false, false); break;
        case 4: RN_SPLIT_LAUNCH(2, 2, false, false, false); break;
        case 5: RN_SPLIT_LAUNCH(2, 2, true, false, false); break;
        default: return RN_EINVAL;
    }
#undef RN_SPLIT_LAUNCH
    RN_LAUNCH_CHECK();
    return RN_OK;
}

int rn_igemm_split_grouped_launch(bool narrow, unsigned tiles, const rn_conv_group *g, const float *w, const float *scale,
                                  const float *shift, hipStream_t s) {
    const bool pre = g->d[0].w_format == 1;
    if (!narrow) {
        int rc = RN_OK;
        if (rn_igemm_mf16_grouped_launch(g, w, scale, shift, s, &rc)) return rc;
    }
    if (g->d[0].w_format == 3) {                             // the fp16 two-term tile (see rn_igemm_split_launch)
        for (int i = 0; i < g->n; ++i)
            if (g->d[i].w_format != 3 || g->d[i].x_amax == nullptr || g->d[i].w_unscale == nullptr) return RN_EINVAL;
        if (narrow) hipLaunchKernelGGL((conv_igemm_split_grouped_kernel<4, 1, 2, 16, 2>), dim3(tiles), dim3(256), 0, s, *g, w, scale, shift);
        else hipLaunchKernelGGL((conv_igemm_split_grouped_kernel<2, 2, 2, 16, 2>), dim3(tiles), dim3(256), 0, s, *g, w, scale, shift);
        RN_LAUNCH_CHECK();
        return RN_OK;
    }
    if (!narrow && split_a_once(&g->d[0])) hipLaunchKernelGGL((conv_igemm_split_grouped_kernel<2, 2, 3>), dim3(tiles), dim3(256), 0, s, *g, w, scale, shift);
    else if (narrow && pre) hipLaunchKernelGGL((conv_igemm_split_grouped_kernel<4, 1, 2>), dim3(tiles), dim3(256), 0, s, *g, w, scale, shift);
    else if (narrow) hipLaunchKernelGGL((conv_igemm_split_grouped_kernel<4, 1, 1>), dim3(tiles), dim3(256), 0, s, *g, w, scale, shift);
    else if (pre) hipLaunchKernelGGL((conv_igemm_split_grouped_kernel<2, 2, 2>), dim3(tiles), dim3(256), 0, s, *g, w, scale, shift);
    else hipLaunchKernelGGL((conv_igemm_split_grouped_kernel<2, 2, 1>), dim3(tiles), dim3(256), 0, s, *g, w, scale, shift);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ---- the pre-split form of a packed weight tensor
__global__ __launch_bounds__(256) void split_weights_kernel(const float *__restrict__ src, void *__restrict__ dst, int64_t chunks) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < chunks) split_store_chunk(src, dst, i);
}

extern "C" int rn_split_weights(const float *w_packed, void *w_split, int64_t rows, int Kpad, void *stream) {
    if (rows <= 0 || Kpad <= 0 || (Kpad & 15)) return RN_EINVAL;
    const int64_t chunks = rows * Kpad / 8;
    hipLaunchKernelGGL(split_weights_kernel, dim3(rn_blocks(chunks, 256)), dim3(256), 0, (hipStream_t)stream, w_packed, w_split, chunks);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ---- RN_FP32_SPLIT3: the fp16 pre-split form (mfma_split.h: split_row_f16, one wave per row)
__global__ __launch_bounds__(256) void split_weights_f16_kernel(const float *__restrict__ src, void *__restrict__ dst,
                                                                 float *__restrict__ unscale, int64_t rows, int Kpad) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < rows) split_row_f16(src, dst, unscale, row, Kpad, threadIdx.x & 63);
}

extern "C" int rn_split_weights_f16(const float *w_packed, void *w_split, float *row_unscale, int64_t rows, int Kpad, void *stream) {
    if (rows <= 0 || Kpad <= 0 || (Kpad & 15) || !row_unscale) return RN_EINVAL;
    hipLaunchKernelGGL(split_weights_f16_kernel, dim3(rn_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream, w_packed, w_split, row_unscale, rows, Kpad);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// The amax words of a tensor nobody left them for (rn_conv_desc.x_amax): one streaming pass, grid.y = image.
__global__ __launch_bounds__(256) void amax_kernel(const float *__restrict__ x, int64_t per_image, void *__restrict__ amax) {
    const float *xi = x + (int64_t)blockIdx.y * per_image;
    const int64_t n4 = per_image >> 2, stride = (int64_t)gridDim.x * 256;
    float am = 0.f;
    if ((((uintptr_t)xi) & 15) == 0) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const float4 q = reinterpret_cast<const float4 *>(xi)[i];
            am = fmaxf(fmaxf(am, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
        }
        if (blockIdx.x == 0 && threadIdx.x < (per_image & 3)) am = fmaxf(am, fabsf(xi[(n4 << 2) + threadIdx.x]));
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_image; i += stride) am = fmaxf(am, fabsf(xi[i]));
    }
    rn_amax_note(amax, blockIdx.y, am);
}
extern "C" int rn_amax(const float *x, int64_t per_image, int n_images, void *amax, void *stream) {
    if (per_image <= 0 || n_images <= 0 || n_images > 65535 || !amax || ((uintptr_t)x & 3)) return RN_EINVAL;
    const int64_t want = (per_image / 4 + 255) / 256;
    const int64_t cap = 2048 / n_images < 8 ? 8 : 2048 / n_images;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)(want < 1 ? 1 : (want > cap ? cap : want)), (unsigned)n_images), dim3(256), 0,
                       (hipStream_t)stream, x, per_image, amax);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

#if RN_STAMP
// diagnostic build: read and clear the K-step stamps (conv_igemm_tile.h)
extern "C" int rn_debug_stamps(unsigned long long *out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rn_stamps), sizeof(rn_stamps)) != hipSuccess) return -1;
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rn_stamps), z, sizeof(z)) != hipSuccess) return -1;
    return RN_OK;
}
#endif
