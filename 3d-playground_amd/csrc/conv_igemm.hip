// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak).
//
// Replaces every nn.Conv2d of the detector -- ResNet stem / BasicBlock / Bottleneck convs (D/model.py:213,
// D/utils.py:6-80), PyramidFeatures (D/model.py:59-117), RegressionModel / ClassificationModel towers and
// outputs (D/model.py:120-205) -- together with what the reference runs as separate kernels right after them:
// the frozen BatchNorm affine (D/model.py:278-282), conv bias, the residual add + ReLU of a block
// (D/utils.py:38-43, 75-80), the FPN nearest-upsample + add with crop (D/model.py:88-108), the head's Sigmoid
// and permute(0,2,3,1)+view (D/model.py:155-157, 196-205).  The same kernel computes data gradients: dgrad is a
// convolution of dY with re-packed weights under a different output->input coordinate map.
//
// GEMM view:  M = N*Ho*Wo output pixels,  N = Cout,  K = kh*kw*Cin.
// Layout:     activations NHWC fp32 (channels contiguous: 16-byte loads along K, 128-byte stores along Cout);
//             weights packed [Cout][kh][kw][Cin] (K contiguous per output channel) by pack kernels.
// Tile:       workgroup 64*WM x 64*WN outputs, K-step 16, four workgroups per CU; 4 waves, each a 64x64 sub-tile = 2x2 MFMA 32x32
//             accumulators (64 VGPRs).  Both operands are staged as K-contiguous rows in LDS, and they get there by
//             direct-to-LDS buffer loads (16 bytes per lane, 1 KiB of whole rows per wave instruction): no staging
//             registers, no ds_write, no store phase.  Rows are unpadded (the load fills LDS linearly), so the 16-byte
//             chunks of a row are XOR-permuted -- on the source address of the load and again on the fragment read
//             ds_read_b128 at [row = lane&31][chunk 2*step + (lane>>5)] -- which keeps those reads conflict-free
//             (conv_igemm_tile.h: lds_swz).  One b128 per operand tile feeds four MFMAs.  Zero padding, rows past M
//             and weight rows past Cout are out-of-range buffer offsets: the hardware writes 0.0 for them.  On the
//             fast path (Cin a multiple of the K-step) a K-step costs the vector ALU nothing outside the MFMA stream:
//             the weight offsets are fixed for the whole kernel, the activation offsets per filter tap, and the K /
//             channel advance rides in the scalar offset.  The loads of step t+1 are issued before the 64 MFMAs of
//             step t, into the other LDS buffer: one barrier per K-step.
// Grid:       1-D, tile id remapped so that consecutive tiles (neighbouring pixel rows, both Cout halves) share
//             an XCD's L2: halo rows and the 9 taps of a 3x3 filter are re-read from L2, not HBM.
//
// Epilogue:   the 128x128 (256x64) accumulator tile is staged through the now idle LDS and leaves as float4 rows:
//             out, residual / gradient addend and ReLU mask are all 16-byte coalesced accesses.
//
// Roofline: MFMA (fp32 157.3 TF).  Per K-step of 16 a wave issues 32 MFMAs (2048 cycles) against 8 ds_read_b128 and
// 4 direct-to-LDS loads.  Small-K layers (1x1, Cin 64..128) are HBM-bound instead:
// e.g. 1x1 64->256 at 270x480 moves 1.33 GB per 34 GFLOP.
#include "conv_igemm_tile.h"

template <int WM, int WN, bool GENERAL, int BK, bool RELU = false, bool RAW = false>
__global__ __launch_bounds__(256, 4) void conv_igemm_kernel(const rn_conv_desc d, const float *__restrict__ x,
                                                            const float *__restrict__ w, float *__restrict__ y,
                                                            const float *__restrict__ scale, const float *__restrict__ shift,
                                                            const float *__restrict__ add, const float *__restrict__ mask,
                                                            const float *__restrict__ add2) {
    conv_igemm_tile<WM, WN, GENERAL, BK, RELU, RAW>(d, x, w, y, scale, shift, add, mask, add2, xcd_remap(blockIdx.x, gridDim.x));
}

// Grouped launch: the workgroup looks up which problem its tile belongs to (wave-uniform compare chain, static
// indices into the by-value table) and runs the same tile code with that problem's descriptor and pointers.
template <int WM, int WN, int BK>
__global__ __launch_bounds__(256, 4) void conv_igemm_grouped_kernel(const rn_conv_group g,
                                                            const float *__restrict__ w, const float *__restrict__ scale,
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
    conv_igemm_tile<WM, WN, true, BK>(d, x, w, y, scale, shift, add, mask, nullptr, tile - first);
}

// conv_igemm_split.hip: the same instances with split-operand products
int rn_igemm_split_launch(int variant, unsigned tiles, const rn_conv_desc *d, const float *x, const float *w, float *y,
                          const float *scale, const float *shift, const float *add, const float *mask, const float *add2,
                          hipStream_t s);
int rn_igemm_split_grouped_launch(bool narrow, unsigned tiles, const rn_conv_group *g, const float *w, const float *scale,
                                  const float *shift, hipStream_t s);

extern "C" int rn_conv_igemm_grouped(const rn_conv_group *g, const float *w_packed, const float *scale, const float *shift,
                                     void *stream) {
    if (g->n < 1 || g->n > RN_MAX_GROUP) return RN_EINVAL;
    const rn_conv_desc &d0 = g->d[0];
    const bool narrow = d0.Cout <= 64;
    int prev = 0;
    for (int i = 0; i < g->n; ++i) {
        const rn_conv_desc &d = g->d[i];
        const int rc = check_desc(&d);
        if (rc) return rc;
        if (d.Cin != d0.Cin || d.Cout != d0.Cout || d.kh != d0.kh || d.kw != d0.kw || d.add2_mode != 0) return RN_EINVAL;
        if (d.in_relu) return RN_EINVAL;                       // the input ReLU has its own kernel: rn_conv_igemm only
        if ((d.add_mode != 0) != (g->add[i] != nullptr) || (d.mask_mode != 0) != (g->mask[i] != nullptr)) return RN_EINVAL;
        const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
        const int64_t tiles = narrow ? (M + 255) / 256 : ((M + 127) / 128) * ((d.Cout + 127) / 128);
        if (g->tile_end[i] - prev != tiles) return RN_EINVAL;
        prev = g->tile_end[i];
    }
    const dim3 grid((unsigned)prev), block(256);
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < g->n; ++i)
        if (g->d[i].w_format != d0.w_format) return RN_EINVAL;
    if (d0.w_format == 2) return RN_EINVAL;                                          // one-term products: single launches only
    const bool split_mode = rn_get_fp32_mfma() != RN_FP32_NATIVE;                    // RN_FP32_SPLIT or RN_FP32_SPLIT3
    if (d0.w_format != 0 && !split_mode) return RN_EINVAL;                           // the pre-split forms are operands of the split kernels only
    if (split_mode && (d0.w_format != 0 || d0.kh * d0.kw * d0.Cin >= rn_fp32_split_min_k()))
        return rn_igemm_split_grouped_launch(narrow, (unsigned)prev, g, w_packed, scale, shift, s);
    if (narrow) hipLaunchKernelGGL((conv_igemm_grouped_kernel<4, 1, 16>), grid, block, 0, s, *g, w_packed, scale, shift);
    else hipLaunchKernelGGL((conv_igemm_grouped_kernel<2, 2, 16>), grid, block, 0, s, *g, w_packed, scale, shift);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_conv_igemm(const rn_conv_desc *d, const float *x, const float *w_packed, float *y, const float *scale,
                             const float *shift, const float *add, const float *mask, const float *add2, void *stream) {
    const int rc = check_desc(d);
    if (rc) return rc;
    if ((d->add_mode != 0) != (add != nullptr)) return RN_EINVAL;
    if ((d->mask_mode != 0) != (mask != nullptr)) return RN_EINVAL;
    if ((d->add2_mode != 0) != (add2 != nullptr)) return RN_EINVAL;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    hipStream_t s = (hipStream_t)stream;
    const bool dense = d->os == 1 && d->oo_h == 0 && d->oo_w == 0 && d->Hy == d->Ho && d->Wy == d->Wo &&
                       d->y_batch_stride == (int64_t)d->Ho * d->Wo * d->Cout && d->add_mode != 2 && d->add2_mode == 0 &&
                       (d->add_mode == 0 || d->add_batch_stride == d->y_batch_stride);
    // 256 x 64 tile for few output channels: no wasted N half.  (The input-ReLU form exists for the 128 x 128 tile only.)
    const bool narrow = d->Cout <= 64 && !d->in_relu;
    const int64_t tiles = narrow ? (M + 255) / 256 : ((M + 127) / 128) * ((d->Cout + 127) / 128);
    if (tiles > 0x7fffffff) return RN_EINVAL;
    const dim3 grid((unsigned)tiles), block(256);
#define RN_LAUNCH_IGEMM(WM, WN, G, K) \
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, G, K>), grid, block, 0, s, *d, x, w_packed, y, scale, shift, add, mask, add2)
    // K-step 16 (34-41 KB of LDS, four workgroups per CU) everywhere: with the operands arriving by direct-to-LDS loads it
    // ties or beats K-step 32 at two workgroups per CU on every layer shape (measured).
    const bool raw = dense && !narrow && !d->in_relu && !scale && !shift && d->add_mode == 0 && d->mask_mode == 0 && d->act == 0 &&
                     d->sign_out == nullptr && d->y_amax == nullptr;
    if (d->w_format == 2) {
        // pre-split weights, products from the first bf16 terms only: whatever the fp32 product mode (it is not fp32 arithmetic),
        // narrow instances only, no input ReLU (rn_igemm_split_launch)
        if (!narrow || d->in_relu) return RN_EINVAL;
        return rn_igemm_split_launch(dense ? 2 : 3, (unsigned)tiles, d, x, w_packed, y, scale, shift, add, mask, add2, s);
    }
    const bool split_mode = rn_get_fp32_mfma() != RN_FP32_NATIVE;                    // RN_FP32_SPLIT or RN_FP32_SPLIT3
    if (d->w_format != 0 && !split_mode) return RN_EINVAL;
    if (split_mode && (d->w_format != 0 || d->kh * d->kw * d->Cin >= rn_fp32_split_min_k()))
        return rn_igemm_split_launch(raw ? 0 : (d->in_relu ? 1 : (narrow ? (dense ? 2 : 3) : (dense ? 4 : 5))), (unsigned)tiles, d, x,
                                     w_packed, y, scale, shift, add, mask, add2, s);
    if (raw) {                                                           // a plain GEMM: the Winograd stage
        hipLaunchKernelGGL((conv_igemm_kernel<2, 2, false, 16, false, true>), grid, block, 0, s, *d, x, w_packed, y, scale, shift, add, mask, add2);
    } else if (d->in_relu) {
        hipLaunchKernelGGL((conv_igemm_kernel<2, 2, true, 16, true>), grid, block, 0, s, *d, x, w_packed, y, scale, shift, add, mask, add2);
    } else if (narrow) {
        if (dense) RN_LAUNCH_IGEMM(4, 1, false, 16); else RN_LAUNCH_IGEMM(4, 1, true, 16);
    } else {
        if (dense) RN_LAUNCH_IGEMM(2, 2, false, 16); else RN_LAUNCH_IGEMM(2, 2, true, 16);
    }
#undef RN_LAUNCH_IGEMM
    RN_LAUNCH_CHECK();
    return RN_OK;
}
