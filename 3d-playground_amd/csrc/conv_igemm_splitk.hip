// Split-K form of the implicit-GEMM convolution (its own translation unit: see conv_igemm_tile.h).
#include "conv_igemm_tile.h"

// ---------------------------------------------------------------------------------------------- split-K
// For launches whose output is only a few tiles while K is long (the P6 / P7 pyramid levels, layer3 / layer4 of the
// 112x112 crop detector): grid.y slices the K loop, each slice writes its raw partial tile to its own slab of a workspace
// (plain stores: no atomics, no zero-fill, the same bits every run) and conv_splitk_finish_kernel adds the slabs in slice
// order and applies the epilogue.
template <int WM, int WN, int BK, bool RELU = false>
__global__ __launch_bounds__(256, 4) void conv_igemm_splitk_kernel(
    const rn_conv_desc d, const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ ws, int steps_per_slice,
    int nks_total) {
    const int ks_lo = blockIdx.y * steps_per_slice;
    int ks_hi = ks_lo + steps_per_slice;
    if (ks_hi > nks_total) ks_hi = nks_total;
    const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
    // y / scale / shift / add / mask / add2 are never touched on the partial path; they get real pointers all the same:
    // literal nullptrs make the (dead) epilogue a store through a constant null after inlining, and hipcc 7.2's
    // optimizer segfaults on that.
    float *slab = ws + (int64_t)blockIdx.y * M * d.Cout;
    conv_igemm_tile<WM, WN, false, BK, RELU>(d, x, w, slab, w, w, x, x, x, (int)blockIdx.x, ks_lo, ks_hi, slab);
}

template <bool GENERAL>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const rn_conv_desc d, const float *__restrict__ ws, int slices,
                                                                 float *__restrict__ y, const float *__restrict__ scale,
                                                                 const float *__restrict__ shift, const float *__restrict__ add,
                                                                 const float *__restrict__ mask, const float *__restrict__ add2) {
    const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
    const int cpr = (d.Cout + 3) / 4;                        // 4-channel chunks per output pixel
    const int64_t chunk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float rn_am = 0.f;                                       // rn_conv_desc.y_amax: what this thread's chunk stores, into its image's word
    const bool rn_span = false;
    const int64_t m_mine = chunk < M * cpr ? chunk / cpr : M - 1;
    const int n_mine = (int)((unsigned)m_mine / (unsigned)(d.Ho * d.Wo));
    if (chunk < M * cpr) {
    const int64_t m = chunk / cpr;
    const int col = (int)(chunk - m * cpr) * 4;
    const bool vec = (d.Cout & 3) == 0;
    const int ncol = vec ? 4 : (d.Cout - col < 4 ? d.Cout - col : 4);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const float *p = ws + m * d.Cout + col;
    for (int sl = 0; sl < slices; ++sl, p += M * d.Cout) {   // slice order: deterministic
        if (vec) { const float4 q = *reinterpret_cast<const float4 *>(p); a[0] += q.x; a[1] += q.y; a[2] += q.z; a[3] += q.w; }
        else for (int j = 0; j < ncol; ++j) a[j] += p[j];
    }
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < ncol && scale != nullptr) sc[j] = scale[col + j];
        if (j < ncol && shift != nullptr) sh[j] = shift[col + j];
    }
    const float4 t = make_float4(a[0], a[1], a[2], a[3]);
    const int HoWo = d.Ho * d.Wo;
    RN_EPI_CHUNK_BODY(GENERAL);
    }
    if (chunk < M * cpr) rn_amax_note(d.y_amax, n_mine, rn_am);
}

// Slices worth using for this problem (1 = do not split) -- few output tiles and a long K loop.
static int splitk_slices(const rn_conv_desc *d) {
    if (!rn_get_option(RN_OPT_SPLITK)) return 1;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    const bool narrow = d->Cout <= 64 && !d->in_relu;      // the input-ReLU form exists for the 128 x 128 tile only
    const int64_t tiles = narrow ? (M + 255) / 256 : ((M + 127) / 128) * ((d->Cout + 127) / 128);
    const int K = d->kh * d->kw * d->Cin;
    const int bk = 16;
    const int nks = ((K + 31) / 32 * 32) / bk;
    if (tiles >= 160 || nks < 32) return 1;                  // K < 512: not worth a second pass over the output
    int64_t s = (512 + tiles - 1) / tiles;                   // ~2 workgroups per CU in total
    if (s > nks / 16) s = nks / 16;                          // at least 16 K-steps (256 of K) per slice
    if (s > 32) s = 32;
    return s < 2 ? 1 : (int)s;
}

extern "C" int64_t rn_conv_splitk_workspace_bytes(const rn_conv_desc *d) {
    if (check_desc(d)) return 0;
    const int s = splitk_slices(d);
    return s <= 1 ? 0 : (int64_t)s * d->N * d->Ho * d->Wo * d->Cout * (int64_t)sizeof(float);
}

extern "C" int rn_conv_igemm_splitk(const rn_conv_desc *d, const float *x, const float *w_packed, float *y, const float *scale,
                                    const float *shift, const float *add, const float *mask, const float *add2,
                                    void *workspace, void *stream) {
    const int rc = check_desc(d);
    if (rc) return rc;
    if (d->w_format != 0) return RN_EINVAL;                  // the pre-split weight form: rn_conv_igemm only
    if ((d->add_mode != 0) != (add != nullptr)) return RN_EINVAL;
    if ((d->mask_mode != 0) != (mask != nullptr)) return RN_EINVAL;
    if ((d->add2_mode != 0) != (add2 != nullptr)) return RN_EINVAL;
    const int slices = splitk_slices(d);
    if (slices <= 1 || !workspace) return RN_EINVAL;         // ask rn_conv_splitk_workspace_bytes first
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    hipStream_t s = (hipStream_t)stream;
    const bool narrow = d->Cout <= 64 && !d->in_relu;
    const int64_t tiles = narrow ? (M + 255) / 256 : ((M + 127) / 128) * ((d->Cout + 127) / 128);
    const int K = d->kh * d->kw * d->Cin;
    const int bk = 16;
    const int nks = ((K + 31) / 32 * 32) / bk;
    const int per = (nks + slices - 1) / slices;
    const int used = (nks + per - 1) / per;                  // slices that actually get K-steps
    const dim3 grid((unsigned)tiles, (unsigned)used), block(256);
    float *ws = reinterpret_cast<float *>(workspace);
    if (d->in_relu) hipLaunchKernelGGL((conv_igemm_splitk_kernel<2, 2, 16, true>), grid, block, 0, s, *d, x, w_packed, ws, per, nks);
    else if (narrow) hipLaunchKernelGGL((conv_igemm_splitk_kernel<4, 1, 16>), grid, block, 0, s, *d, x, w_packed, ws, per, nks);
    else hipLaunchKernelGGL((conv_igemm_splitk_kernel<2, 2, 16>), grid, block, 0, s, *d, x, w_packed, ws, per, nks);
    RN_LAUNCH_CHECK();
    const bool dense = d->os == 1 && d->oo_h == 0 && d->oo_w == 0 && d->Hy == d->Ho && d->Wy == d->Wo &&
                       d->y_batch_stride == (int64_t)d->Ho * d->Wo * d->Cout && d->add_mode != 2 && d->add2_mode == 0 &&
                       (d->add_mode == 0 || d->add_batch_stride == d->y_batch_stride);
    const int64_t chunks = M * ((d->Cout + 3) / 4);
    const dim3 fgrid((unsigned)((chunks + 255) / 256));
    if (dense) hipLaunchKernelGGL((conv_splitk_finish_kernel<false>), fgrid, block, 0, s, *d, (const float *)ws, used, y, scale, shift, add, mask, add2);
    else hipLaunchKernelGGL((conv_splitk_finish_kernel<true>), fgrid, block, 0, s, *d, (const float *)ws, used, y, scale, shift, add, mask, add2);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

