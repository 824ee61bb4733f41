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
// Tile:       workgroup 64*WM x 64*WN outputs, K-step 32; 4 waves, each a 64x64 sub-tile = 2x2 MFMA 32x32
//             accumulators (64 VGPRs).  Both operands are staged as K-contiguous rows in LDS with a 4-float pad
//             (row stride 144 B): ds_write_b128 by rows is conflict-free, and the fragment read
//             ds_read_b128 at [row = lane&31][k = 8*step + 4*(lane>>5) .. +3] is conflict-free as well
//             (16-lane groups hit 16 distinct 16-byte slots: 9*row mod 16 is a bijection on the group's rows);
//             one b128 per operand tile feeds four MFMAs.  Global->register loads run two K-steps ahead (two register
//             stages at K-step 32): issued before the 64 MFMAs of step t, written to the idle LDS buffer one step later,
//             after the MFMAs: one barrier per K-step.  They are unconditional, from clamped addresses; zero-fill and the
//             input ReLU happen on the way into LDS (conv_igemm_tile.h).
// Grid:       1-D, tile id remapped so that consecutive tiles (neighbouring pixel rows, both Cout halves) share
//             an XCD's L2: halo rows and the 9 taps of a 3x3 filter are re-read from L2, not HBM.
//
// Epilogue:   the 128x128 (256x64) accumulator tile is staged through the now idle LDS and leaves as float4 rows:
//             out, residual / gradient addend and ReLU mask are all 16-byte coalesced accesses.
//
// Roofline: MFMA (fp32 157.3 TF).  Per K-step a wave issues 64 MFMAs (4096 cycles) against 16 ds_read_b128,
// 8 global_load_dwordx4 and 8 ds_write_b128.  Small-K layers (1x1, Cin 64..128) are HBM-bound instead:
// e.g. 1x1 64->256 at 270x480 moves 1.33 GB per 34 GFLOP.
#include "conv_igemm_tile.h"

#define RN_DEFAULT_WP 0          // 1: barrier-free wave-private variant for the 128x128 tile

template <int WM, int WN, bool GENERAL, int BK>
__global__ __launch_bounds__(256, (BK == 16 && WM == 2) ? 3 : 2) void conv_igemm_kernel(const rn_conv_desc d, const float *__restrict__ x,
                                                            const float *__restrict__ w, float *__restrict__ y,
                                                            const float *__restrict__ scale, const float *__restrict__ shift,
                                                            const float *__restrict__ add, const float *__restrict__ mask,
                                                            const float *__restrict__ add2) {
    conv_igemm_tile<WM, WN, GENERAL, BK>(d, x, w, y, scale, shift, add, mask, add2, xcd_remap(blockIdx.x, gridDim.x));
}

// Grouped launch: the workgroup looks up which problem its tile belongs to (wave-uniform compare chain, static
// indices into the by-value table) and runs the same tile code with that problem's descriptor and pointers.
template <int WM, int WN, int BK>
__global__ __launch_bounds__(256, (BK == 16 && WM == 2) ? 3 : 2) void conv_igemm_grouped_kernel(const rn_conv_group g,
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

// ------------------------------------------------------------------------------------------------
// Barrier-free variant: wave-private operand slabs (see the K loop).  Same tile, same epilogue.
template <int WM, int WN, bool GENERAL>
__global__ __launch_bounds__(256, 2) void conv_igemm_wp_kernel(const rn_conv_desc d, const float *__restrict__ x,
                                                            const float *__restrict__ w, float *__restrict__ y,
                                                            const float *__restrict__ scale, const float *__restrict__ shift,
                                                            const float *__restrict__ add, const float *__restrict__ mask,
                                                            const float *__restrict__ add2) {
    constexpr int BK = 32;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LDK = BK + 4;                            // padded LDS row, floats (conflict-free b128 reads)
    constexpr int CPK = BK / 4;                            // 16-byte chunks per staged row
    constexpr int RPS = 64 / CPK;                          // rows one WAVE stages per pass (8)
    constexpr int AR = 64 / RPS, BR = 64 / RPS;            // 8 + 8 float4 per lane per K-step: the wave's own 64 A and 64 B rows
    static_assert(WM * WN == 4, "4 waves");
    __shared__ float lds[2][(BM + BN) * LDK];              // used as 4 wave-private [128][LDK] slabs in the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (d.Cout + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int HoWo = d.Ho * d.Wo;
    const int64_t M = (int64_t)d.N * HoWo;
    const int K = d.kh * d.kw * d.Cin;
    const int Kpad = (K + 31) / 32 * 32;                   // packed weight rows are zero-padded to a multiple of 32
    const int nks = Kpad / BK;
    const int dmask = (1 << d.div_shift) - 1;

    // ---- per-thread staging geometry: chunk column q (4 floats of K), rows srow + 32*i
    const int q = lane % CPK, srow = lane / CPK;
    const float *a_base[AR];
    int a_h[AR], a_w[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int64_t m = (int64_t)m0 + wm * 64 + srow + RPS * i;
        if (m < M) {
            const int n = (int)(m / HoWo);
            const int rem = (int)(m - (int64_t)n * HoWo);
            const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
            a_base[i] = x + (int64_t)n * d.x_batch_stride;
            a_h[i] = oh * d.a + d.p;
            a_w[i] = ow * d.a + d.p_w;
        } else {
            a_base[i] = x;
            a_h[i] = -(1 << 28);                           // fails every bounds test
            a_w[i] = 0;
        }
    }
    const float *b_base[BR];
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + wn * 64 + srow + RPS * i;
        b_ok[i] = n < d.Cout;
        b_base[i] = w + (int64_t)(b_ok[i] ? n : 0) * Kpad + 4 * q;
    }

    float4 ra[AR], rb[BR];
    // Fast path (Cin a multiple of the K-step, i.e. every layer but the 4-channel stem and channel-padded head
    // gradients): a K-step lies inside one filter tap, so the per-row input coordinate, bounds test and pixel offset
    // are recomputed only when the tap changes (every Cin/BK steps); in between a step is one add per row.
    const bool fast = (d.Cin % BK) == 0;
    int a_pix[AR];                                         // fast path: in-image offset of the row's pixel for the tap, <0 = zero
    int f_r = 0, f_s = 0, f_c = 0;                         // tap (r,s) and channel offset of the NEXT step to load
    auto load_step = [&](int ks) {
        if (fast) {
            if (f_c == 0) {                                // new tap (wave-uniform)
                const bool tap_ok = f_r < d.kh;
                const int hoff = f_r * d.b, woff = f_s * d.b;
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const int nh = a_h[i] + hoff, nw = a_w[i] + woff;
                    const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                    const bool ok = tap_ok && ((nh | nw) >= 0) && (((nh | nw) & dmask) == 0) && ih < d.Hi && iw < d.Wi;
                    a_pix[i] = ok ? (ih * d.Wi + iw) * d.Cin : -1;
                }
            }
            const int c = f_c + 4 * q;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                float4 v = a_pix[i] >= 0 ? *reinterpret_cast<const float4 *>(a_base[i] + (a_pix[i] + c))
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                if (d.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                ra[i] = v;
            }
            f_c += BK;
            if (f_c >= d.Cin) { f_c = 0; if (++f_s == d.kw) { f_s = 0; ++f_r; } }
        } else {
            const int k = ks * BK + 4 * q;
            const int tap = k / d.Cin;
            const int c0 = k - tap * d.Cin;
            const int r = tap / d.kw, s = tap - r * d.kw;
            const bool tap_ok = r < d.kh;
            const int hoff = r * d.b, woff = s * d.b;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int nh = a_h[i] + hoff, nw = a_w[i] + woff;
                const int ih = nh >> d.div_shift, iw = nw >> d.div_shift;
                const bool ok = tap_ok && ((nh | nw) >= 0) && (((nh | nw) & dmask) == 0) && ih < d.Hi && iw < d.Wi;
                // offsets inside one image fit 32 bits (x_batch_stride < 2^31 floats is checked by the launcher)
                float4 v = ok ? *reinterpret_cast<const float4 *>(a_base[i] + ((ih * d.Wi + iw) * d.Cin + c0))
                              : make_float4(0.f, 0.f, 0.f, 0.f);
                if (d.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            rb[i] = b_ok[i] ? *reinterpret_cast<const float4 *>(b_base[i] + ks * BK) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float *slab = &lds[0][0] + wave * (128 * LDK);         // this wave's [64 A rows | 64 B rows][LDK]
    auto store_step = [&](int) {
        float *A = slab, *B = slab + 64 * LDK;
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<float4 *>(A + (srow + RPS * i) * LDK + 4 * q) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *reinterpret_cast<float4 *>(B + (srow + RPS * i) * LDK + 4 * q) = rb[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // No workgroup barrier in the K loop: every wave owns its operands (its 64 A rows and 64 B rows are loaded by the
    // wave itself, so a tile's rows are fetched by two waves -- L1/L2 absorb the duplicate), and LDS operations of one
    // wave execute in order, so "write slab, read fragments" needs only the compiler kept from reordering them.
    load_step(0);
    store_step(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int frag = (lane & 31) * LDK + (lane >> 5) * 4;   // [row = lane&31][k = 4*(lane>>5)]
    for (int ks = 0; ks < nks; ++ks) {
        if (ks + 1 < nks) load_step(ks + 1);
        const float *A = slab + frag;
        const float *B = slab + 64 * LDK + frag;
        float4 fa[BK / 8][2], fb[BK / 8][2];
#pragma unroll
        for (int st = 0; st < BK / 8; ++st) {                // all fragments of the step first: the slab is free afterwards
            fa[st][0] = *reinterpret_cast<const float4 *>(A + st * 8);
            fa[st][1] = *reinterpret_cast<const float4 *>(A + 32 * LDK + st * 8);
            fb[st][0] = *reinterpret_cast<const float4 *>(B + st * 8);
            fb[st][1] = *reinterpret_cast<const float4 *>(B + 32 * LDK + st * 8);
        }
#pragma unroll
        for (int st = 0; st < BK / 8; ++st) {
            const float av[2][4] = {{fa[st][0].x, fa[st][0].y, fa[st][0].z, fa[st][0].w}, {fa[st][1].x, fa[st][1].y, fa[st][1].z, fa[st][1].w}};
            const float bv[2][4] = {{fb[st][0].x, fb[st][0].y, fb[st][0].z, fb[st][0].w}, {fb[st][1].x, fb[st][1].y, fb[st][1].z, fb[st][1].w}};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm][j], bv[tn][j], acc[tm][tn], 0, 0, 0);
        }
        if (ks + 1 < nks) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            store_step(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();                                         // the epilogue re-uses the whole LDS as one tile

    // ---- epilogue: v = scale[c]*acc + shift[c]; [mask before add]; v += add (+ add2); act; [mask after]
    // The accumulator tile goes through LDS (the staging buffers are free after the last barrier) so that global
    // memory sees 16-byte accesses, 32 consecutive lanes on one 512-byte row segment: out, add and mask all move as
    // float4.  Accumulator element e of lane l is row (e&3) + 8*(e>>2) + 4*(l>>5), column l&31 of its 32x32 tile.
    constexpr int LDT = BN + 4;
    constexpr int EP = (BM * LDT > 2 * (BM + BN) * LDK) ? 2 : 1;   // passes when the tile outgrows the staging LDS
    constexpr int RP = BM / EP;                                    // tile rows per pass
    static_assert(RP * LDT <= 2 * (BM + BN) * LDK && RP % 64 == 0, "output tile pass must fit the staging LDS");
    float *T = &lds[0][0];
    constexpr int CPR = BN / 4, RPP = 256 / CPR;             // 16-byte chunks per tile row, rows per pass of stores
    const int c4 = tid % CPR;
    const int col = n0 + 4 * c4;
    const bool col_ok = col < d.Cout;
    const bool vec = (d.Cout & 3) == 0;                      // then col+3 < Cout and every row offset is 16-byte aligned
    const int ncol = vec ? 4 : (d.Cout - col < 4 ? d.Cout - col : 4);
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (col_ok && j < ncol && scale != nullptr) sc[j] = scale[col + j];
        if (col_ok && j < ncol && shift != nullptr) sh[j] = shift[col + j];
    }
#pragma unroll
    for (int pass = 0; pass < EP; ++pass) {
        if (pass) __syncthreads();
        if ((wm * 64) / RP == pass) {
            const int rbase = wm * 64 - pass * RP;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        T[(rbase + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * LDT + wn * 64 + tn * 32 + (lane & 31)] =
                            acc[tm][tn][e];
        }
        __syncthreads();
        for (int r = tid / CPR; r < RP; r += RPP) {
            const int64_t m = (int64_t)m0 + pass * RP + r;
            if (m >= M || !col_ok) break;
            const float4 t = *reinterpret_cast<const float4 *>(T + r * LDT + 4 * c4);
            float v[4] = {t.x * sc[0] + sh[0], t.y * sc[1] + sh[1], t.z * sc[2] + sh[2], t.w * sc[3] + sh[3]};
            int64_t off, aoff = -1, a2off = -1;
            if (!GENERAL) {
                off = m * d.Cout + col;
                if (d.add_mode == 1) aoff = off;
            } else {
                const int n = (int)(m / HoWo);
                const int rem = (int)(m - (int64_t)n * HoWo);
                const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
                const int ph = oh * d.os + d.oo_h, pw = ow * d.os + d.oo_w;
                const int64_t pix = (int64_t)ph * d.Wy + pw;
                off = (int64_t)n * d.y_batch_stride + pix * d.Cout + col;
                if (d.add_mode == 1) aoff = (int64_t)n * d.add_batch_stride + pix * d.Cout + col;
                else if (d.add_mode == 2)                    // nearest x2 upsample of [N,Ha,Wa,Cout], cropped (D/model.py:88-108)
                    aoff = (int64_t)n * d.add_batch_stride + ((int64_t)(oh >> 1) * d.Wa + (ow >> 1)) * d.Cout + col;
                if (d.add2_mode == 3 && ((ph | pw) & 1) == 0)
                    a2off = (int64_t)n * d.add2_batch_stride + ((int64_t)(ph >> 1) * d.Wa2 + (pw >> 1)) * d.Cout + col;
            }
            float mk[4] = {1.f, 1.f, 1.f, 1.f}, ad[4] = {0.f, 0.f, 0.f, 0.f};
            if (vec) {
                if (d.mask_mode != 0) { const float4 q = *reinterpret_cast<const float4 *>(mask + off); mk[0] = q.x; mk[1] = q.y; mk[2] = q.z; mk[3] = q.w; }
                if (aoff >= 0) { const float4 q = *reinterpret_cast<const float4 *>(add + aoff); ad[0] = q.x; ad[1] = q.y; ad[2] = q.z; ad[3] = q.w; }
                if (a2off >= 0) { const float4 q = *reinterpret_cast<const float4 *>(add2 + a2off); ad[0] += q.x; ad[1] += q.y; ad[2] += q.z; ad[3] += q.w; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < ncol && d.mask_mode != 0) mk[j] = mask[off + j];
                    if (j < ncol && aoff >= 0) ad[j] = add[aoff + j];
                    if (j < ncol && a2off >= 0) ad[j] += add2[a2off + j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = v[j];
                if (d.mask_mode == 1) u = mk[j] > 0.f ? u : 0.f;
                u += ad[j];
                if (d.act == 1) u = fmaxf(u, 0.f);
                else if (d.act == 2) u = 1.0f / (1.0f + expf(-u));
                if (d.mask_mode == 2) u = mk[j] > 0.f ? u : 0.f;
                v[j] = u;
            }
            if (vec) {
                *reinterpret_cast<float4 *>(y + off) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < ncol) y[off + j] = v[j];
            }
        }
    }
}

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
        if ((d.add_mode != 0) != (g->add[i] != nullptr) || (d.mask_mode != 0) != (g->mask[i] != nullptr)) return RN_EINVAL;
        const int64_t M = (int64_t)d.N * d.Ho * d.Wo;
        const int64_t tiles = narrow ? (M + 255) / 256 : ((M + 127) / 128) * ((d.Cout + 127) / 128);
        if (g->tile_end[i] - prev != tiles) return RN_EINVAL;
        prev = g->tile_end[i];
    }
    const dim3 grid((unsigned)prev), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool bk16 = d0.kh * d0.kw * d0.Cin <= 256;
    if (narrow) hipLaunchKernelGGL((conv_igemm_grouped_kernel<4, 1, 16>), grid, block, 0, s, *g, w_packed, scale, shift);
    else if (bk16) hipLaunchKernelGGL((conv_igemm_grouped_kernel<2, 2, 16>), grid, block, 0, s, *g, w_packed, scale, shift);
    else hipLaunchKernelGGL((conv_igemm_grouped_kernel<2, 2, 32>), grid, block, 0, s, *g, w_packed, scale, shift);
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
    const bool narrow = d->Cout <= 64;                                    // 256 x 64 tile: no wasted N half
    const int64_t tiles = narrow ? (M + 255) / 256 : ((M + 127) / 128) * ((d->Cout + 127) / 128);
    if (tiles > 0x7fffffff) return RN_EINVAL;
    const dim3 grid((unsigned)tiles), block(256);
#define RN_LAUNCH_IGEMM(WM, WN, G, K) \
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, G, K>), grid, block, 0, s, *d, x, w_packed, y, scale, shift, add, mask, add2)
    // K-step 16 keeps 3 workgroups per CU: measured better only when the K loop is a handful of steps long
    static const int bk16_env = getenv("RN_IGEMM_BK16") ? atoi(getenv("RN_IGEMM_BK16")) : -1;
    const int bk16 = bk16_env >= 0 ? bk16_env : (d->kh * d->kw * d->Cin <= 256);
    static const int wp_env = getenv("RN_IGEMM_WP") ? atoi(getenv("RN_IGEMM_WP")) : RN_DEFAULT_WP;
    if (!narrow && wp_env && !bk16) {
        if (dense)
            hipLaunchKernelGGL((conv_igemm_wp_kernel<2, 2, false>), grid, block, 0, s, *d, x, w_packed, y, scale, shift, add, mask, add2);
        else
            hipLaunchKernelGGL((conv_igemm_wp_kernel<2, 2, true>), grid, block, 0, s, *d, x, w_packed, y, scale, shift, add, mask, add2);
    } else if (narrow) {                                                  // K-step 16: 51 KB of LDS, two workgroups per CU
        if (dense) RN_LAUNCH_IGEMM(4, 1, false, 16); else RN_LAUNCH_IGEMM(4, 1, true, 16);
    } else if (bk16) {
        if (dense) RN_LAUNCH_IGEMM(2, 2, false, 16); else RN_LAUNCH_IGEMM(2, 2, true, 16);
    } else {
        if (dense) RN_LAUNCH_IGEMM(2, 2, false, 32); else RN_LAUNCH_IGEMM(2, 2, true, 32);
    }
#undef RN_LAUNCH_IGEMM
    RN_LAUNCH_CHECK();
    return RN_OK;
}
