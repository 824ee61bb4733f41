// Fused anchor<->label IoU matching + focal classification loss + smooth-L1 box loss + vanishing-point loss.
//
// Replaces FocalLoss.forward of the reference -- directional 3D variant D/losses.py:27-362 and 2D variant
// R/losses.py:27-177 -- together with calc_iou (D/losses.py:5-22).  The reference walks the batch in a Python
// loop and runs ~40 small torch kernels per image, materialising an [A,N] IoU matrix, an [A,C] target matrix
// and several [A,C] temporaries (348 ms forward at B=8 on 8 CPU threads).
//
// One launch covers the whole batch.  A workgroup owns a tile of 1024 consecutive anchors of one image:
//   phase 0  the image's label rows are reduced to (box, area, class) in LDS; for the directional variant the
//            box is the min/max envelope of the 8 corners (D/losses.py:93-107), NOT label cols 16:20;
//   phase 1  one lane per anchor; labels whose box misses the bounding box of the wave's 64 consecutive anchors are
//            skipped by a scalar branch (their IoU is exactly 0), the others get the full IoU (fp32, one rounding per operation in the
//            reference's order -- this file is compiled with -ffp-contract=off so the 0.4 / 0.5 band
//            comparisons and the first-maximum argmax are bit-identical to torch CPU), state into LDS;
//            positive anchors (~0.07 %) evaluate the regression / VP terms (forward) or their gradient
//            (backward) on the spot;
//   phase 2  the tile's TILE*C classification values are streamed as float4 (16 B per lane, consecutive lanes
//            consecutive addresses) against the states in LDS; forward accumulates the focal sum, backward
//            writes dcls.
// Sums leave the workgroup as one 4-float partial per (image, tile); a finalize kernel (one wave per image) adds
// them in fp64 in a fixed order (no float atomics anywhere; the classification sum is bit-reproducible, the
// positives' terms up to the order in which a tile queues its handful of positive anchors).
//
// Roofline: HBM.  Forward algorithmic bytes = B*A*C*4 (cls) + A*16 (anchors, L2/MALL-resident after the first
// image) + B*N*cols*4 = 105.9 MB at B=8, A=389 205, C=8.  Backward adds the dcls and dreg
// writes (99.6 + 149.5 MB).
#include <math.h>

#include "common.h"

#define NTHR 256          // threads per workgroup
#define APT 4             // anchors per thread
#define TILE (NTHR * APT) // 1024 anchors per workgroup: label setup amortised, 8 float4 in flight per lane in phase 2
#define NWAVES (NTHR / 64)

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float uniform_f(float v) {      // tell the compiler the value is wave-uniform (scalar branch)
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

struct ImageStats {       // one per image, written by finalize, read by backward
    float npos;           // number of positive anchors
    float has_labels;     // 1 if the image has at least one valid label row
    float cls_scale;      // d cls_loss / d (sum of focal terms of this image)
    float reg_scale;      // d reg_loss / d (sum of smooth-L1 terms of this image)
    float vp_scale;       // d vp_loss  / d (sum over positives of sum_k (1-cos_k))
    float pad[3];
};

static_assert(sizeof(ImageStats) == 32, "layout");

extern "C" int64_t rn_focal_workspace_bytes(int B, int64_t A) {
    const int64_t tiles = (A + TILE - 1) / TILE;
    // [B] statistics | [B] statistics scaled by the incoming gradients (bwd) | [B][tiles] float4 partial sums
    return 2 * (int64_t)B * sizeof(ImageStats) + (int64_t)B * tiles * 4 * sizeof(float);
}

// ----------------------------------------------------------------------------------------------------------
struct LabelLds {
    float x1[RN_MAX_GT], y1[RN_MAX_GT], x2[RN_MAX_GT], y2[RN_MAX_GT], area[RN_MAX_GT];
    int row[RN_MAX_GT];     // original row index in ann[j]
    int count;
};

// Collect valid rows (class column != -1) in their original order.  Serial over N <= 256 rows by thread 0
// would do, but a ballot-free ordered compaction by one wave keeps it short: wave 0 scans rows in chunks of 64.
template <bool DIR>
__device__ __forceinline__ void load_labels(const float *__restrict__ ann_j, int N, LabelLds &L) {
    constexpr int COLS = DIR ? 27 : 5;
    constexpr int CLS_COL = DIR ? 20 : 4;
    if (threadIdx.x < 64) {
        int base = 0;
        for (int r0 = 0; r0 < N; r0 += 64) {
            const int r = r0 + threadIdx.x;
            bool valid = false;
            float bx1 = 0.f, by1 = 0.f, bx2 = 0.f, by2 = 0.f;
            if (r < N) {
                const float *p = ann_j + (int64_t)r * COLS;
                valid = p[CLS_COL] != -1.0f;                                   // D/losses.py:54, R/losses.py:47
                if (DIR) {                                                     // D/losses.py:93-107
                    bx1 = bx2 = p[0];
                    by1 = by2 = p[1];
#pragma unroll
                    for (int k = 1; k < 8; ++k) {
                        bx1 = fminf(bx1, p[2 * k]);
                        bx2 = fmaxf(bx2, p[2 * k]);
                        by1 = fminf(by1, p[2 * k + 1]);
                        by2 = fmaxf(by2, p[2 * k + 1]);
                    }
                } else {
                    bx1 = p[0]; by1 = p[1]; bx2 = p[2]; by2 = p[3];
                }
            }
            const unsigned long long m = __ballot(valid);
            if (valid) {
                const int pos = base + __popcll(m & ((1ull << threadIdx.x) - 1ull));
                L.x1[pos] = bx1; L.y1[pos] = by1; L.x2[pos] = bx2; L.y2[pos] = by2;
                L.area[pos] = (bx2 - bx1) * (by2 - by1);                       // D/losses.py:6
                L.row[pos] = r;
            }
            base += __popcll(m);
        }
        if (threadIdx.x == 0) L.count = base;
    }
    __syncthreads();
}

// IoU max / first argmax of one anchor against the LDS label set.  Exactly calc_iou's operation order
// (D/losses.py:5-22) followed by torch.max(dim=1) (first maximum wins).
__device__ __forceinline__ void match_anchor(const float4 a, const LabelLds &L, float &best, int &arg) {
    const float area_a = (a.z - a.x) * (a.w - a.y);
    best = -1.0f;
    arg = 0;
    for (int n = 0; n < L.count; ++n) {
        float iw = fminf(a.z, L.x2[n]) - fmaxf(a.x, L.x1[n]);
        float ih = fminf(a.w, L.y2[n]) - fmaxf(a.y, L.y1[n]);
        iw = fmaxf(iw, 0.f);
        ih = fmaxf(ih, 0.f);
        const float inter = iw * ih;
        float iou = 0.f;                                     // 0 / max(ua, 1e-8) == 0 exactly: ~99 % of pairs skip the divide
        if (inter > 0.f) {
            float ua = (area_a + L.area[n]) - inter;
            ua = fmaxf(ua, 1e-8f);
            iou = inter / ua;
        }
        if (iou > best) { best = iou; arg = n; }
    }
}

// sign tables of the corner synthesis (D/losses.py:310-327): corner j = c + sl*l + sw*w + sh*h
__device__ __constant__ float kSL[8] = {-1, -1, +1, +1, -1, -1, +1, +1};
__device__ __constant__ float kSW[8] = {-1, +1, -1, +1, -1, +1, -1, +1};
__device__ __constant__ float kSH[8] = {+1, +1, +1, +1, -1, -1, -1, -1};

__device__ __forceinline__ float sl1(float d) {                 // D/losses.py:345-349
    return d <= (1.0f / 9.0f) ? 0.5f * 9.0f * (d * d) : d - 0.5f / 9.0f;
}
__device__ __forceinline__ float sl1_grad(float d) { return d <= (1.0f / 9.0f) ? 9.0f * d : 1.0f; }
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }

// Directional positive anchor: regression (20 values) + VP terms.  BWD writes the 12 regression gradients.
template <bool BWD>
__device__ __forceinline__ void positive_dir(const float4 a, const float *__restrict__ g /*label row*/,
                                             const float *__restrict__ r /*12*/, float &reg_sum, float &vp_sum,
                                             float reg_scale, float vp_scale, float *__restrict__ dr) {
    const float aw = a.z - a.x, ah = a.w - a.y;                               // D/losses.py:37-40
    const float acx = a.x + 0.5f * aw, acy = a.y + 0.5f * ah;
    float rr[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) rr[i] = r[i];
    float t[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) t[i] = g[i];
    float grad[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) grad[i] = 0.f;

    // ---- vanishing-point direction term on RAW pixel targets (D/losses.py:217-304)
    // k=0: back - front; k=1: right - left; k=2: bottom - top   (corner order fbl fbr bbl bbr ftl ftr btl btr)
    const int plus[3][4] = {{2, 3, 6, 7}, {1, 3, 5, 7}, {0, 1, 2, 3}};
    const int minus[3][4] = {{0, 1, 4, 5}, {0, 2, 4, 6}, {4, 5, 6, 7}};
    float vp = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float tv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float sp = ((t[2 * plus[k][0] + c] + t[2 * plus[k][1] + c]) + t[2 * plus[k][2] + c]) + t[2 * plus[k][3] + c];
            const float sm = ((t[2 * minus[k][0] + c] + t[2 * minus[k][1] + c]) + t[2 * minus[k][2] + c]) + t[2 * minus[k][3] + c];
            tv[c] = (sp - sm) / 4.0f;
        }
        const float vx = rr[2 + 2 * k], vy = rr[3 + 2 * k];
        const float rn = sqrtf(vx * vx + vy * vy);
        const float tn = sqrtf(tv[0] * tv[0] + tv[1] * tv[1]);
        const float dot = vx * tv[0] + vy * tv[1];
        const float den = rn * tn;
        const float cosv = dot / den;
        vp += 1.0f - cosv;
        if (BWD) {
            // d(1-cos)/dv = -( t/den - dot * v / (rn^3 * tn) )
            const float k2 = dot / (den * rn * rn);
            grad[2 + 2 * k] += -(tv[0] / den - k2 * vx) * vp_scale;
            grad[3 + 2 * k] += -(tv[1] / den - k2 * vy) * vp_scale;
        }
    }
    vp_sum += vp;

    // ---- smooth-L1 on 20 anchor-normalised values (D/losses.py:310-350)
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float pred = ((rr[c] + kSL[j] * rr[2 + c]) + kSW[j] * rr[4 + c]) + kSH[j] * rr[6 + c];
            const float tn = c == 0 ? (t[2 * j] - acx) / aw : (t[2 * j + 1] - acy) / ah;
            const float w = j >= 4 ? 0.5f : 1.0f;                             // D/losses.py:343 (cols 8..15)
            const float e = tn - pred;
            const float d = fabsf(e) * w;
            acc += sl1(d);
            if (BWD) {
                const float gp = -sgn(e) * w * sl1_grad(d) * reg_scale;
                grad[c] += gp;
                grad[2 + c] += kSL[j] * gp;
                grad[4 + c] += kSW[j] * gp;
                grad[6 + c] += kSH[j] * gp;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                             // 2D box part, cols 16..19
        const float tn = (k & 1) == 0 ? (t[16 + k] - acx) / aw : (t[16 + k] - acy) / ah;
        const float e = tn - rr[8 + k];
        const float d = fabsf(e);
        acc += sl1(d);
        if (BWD) grad[8 + k] += -sgn(e) * sl1_grad(d) * reg_scale;
    }
    reg_sum += acc;
    if (BWD) {
#pragma unroll
        for (int i = 0; i < 12; ++i) dr[i] = grad[i];
    }
}

// 2D positive anchor (R/losses.py:129-168).
template <bool BWD>
__device__ __forceinline__ void positive_2d(const float4 a, const float *__restrict__ g, const float *__restrict__ r,
                                            float &reg_sum, float reg_scale, float *__restrict__ dr) {
    const float aw = a.z - a.x, ah = a.w - a.y;
    const float acx = a.x + 0.5f * aw, acy = a.y + 0.5f * ah;
    float gw = g[2] - g[0], gh = g[3] - g[1];
    const float gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
    gw = fmaxf(gw, 1.0f);                                                     // R/losses.py:143-144
    gh = fmaxf(gh, 1.0f);
    float t[4];
    t[0] = ((gcx - acx) / aw) / 0.1f;                                         // R/losses.py:146-157
    t[1] = ((gcy - acy) / ah) / 0.1f;
    t[2] = logf(gw / aw) / 0.2f;
    t[3] = logf(gh / ah) / 0.2f;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float e = t[k] - r[k];
        const float d = fabsf(e);
        acc += sl1(d);
        if (BWD) dr[k] = -sgn(e) * sl1_grad(d) * reg_scale;
    }
    reg_sum += acc;
}

// Focal term / gradient of one classification value, branch-free and lean (the kernel is VALU-issue bound:
// ~11 instructions per value).  With a = (pos ? p : 1-p), m = 1-a:
//   loss     = w * m^2 * (-ln a)                   (D/losses.py:137-146: both branches of the reference's where())
//   dloss/dp = +-w * (2 m ln a - m^2 / a)          (+ for the positive class, - otherwise); 0 outside the clamp range
// `wl` carries the weight: forward  w*ln2 (0 for an ignored anchor) since ln a = ln2 * log2 a (v_log_f32, ~1 ulp);
//                          backward +-w*scale (0 for an ignored anchor).  v_rcp_f32 (~1 ulp) replaces the divide.
#define RN_LN2 0.69314718055994530942f
template <bool BWD>
__device__ __forceinline__ float focal_elem(float x, bool pos, float wl) {
    const float p = __builtin_amdgcn_fmed3f(x, 1e-4f, 1.0f - 1e-4f);          // clamp, D/losses.py:56
    const float a = pos ? p : 1.0f - p;
    const float m = 1.0f - a;
    const float l2 = __builtin_amdgcn_logf(a);                                // log2(a), a in [1e-4, 1): no denormals
    if (!BWD) return (wl * (m * m)) * (-l2);
    const float g = wl * ((2.0f * RN_LN2) * (m * l2) - (m * m) * __builtin_amdgcn_rcpf(a));
    return (x < 1e-4f || x > 1.0f - 1e-4f) ? 0.f : g;                         // clamp passes no gradient outside
}

// ----------------------------------------------------------------------------------------------------------
template <bool DIR, bool BWD>
__global__ __launch_bounds__(NTHR) void focal_kernel(const float *__restrict__ cls, const float *__restrict__ reg,
                                                     const float4 *__restrict__ anchors, const float *__restrict__ ann,
                                                     int64_t A, int C, int N, float *__restrict__ partials,
                                                     const ImageStats *__restrict__ stats, float *__restrict__ dcls,
                                                     float *__restrict__ dreg) {
    constexpr int COLS = DIR ? 27 : 5;
    constexpr int CLS_COL = DIR ? 20 : 4;
    constexpr int NREG = DIR ? 12 : 4;
    __shared__ LabelLds L;
    __shared__ int s_state[TILE];     // -1 ignore, 0 negative, 1 + class for a positive
    __shared__ int s_pos[TILE];       // queue of positive anchors of the tile: local index | label row << 16
    __shared__ int s_npos;
    __shared__ float s_red[NWAVES * 4];

    const int j = blockIdx.y;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const float *ann_j = ann + (int64_t)j * N * COLS;
    if (threadIdx.x == 0) s_npos = 0;
    // Streaming-phase loads are issued first: their HBM latency hides behind label setup and assignment.
    const int64_t nA = (A - tile0 < TILE) ? (A - tile0) : TILE;               // anchors in this tile
    const int64_t elems = nA * C;
    const int64_t base = ((int64_t)j * A + tile0) * C;
    const float *src = cls + base;
    constexpr int PRE = 8;                                                     // float4 per lane held in registers
    const bool pre = (C & 3) == 0 && C <= PRE;                                 // then a tile is <= PRE*NTHR float4
    float4 xv[PRE];
    if (pre) {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        const int n4 = (int)(elems >> 2);
#pragma unroll
        for (int k = 0; k < PRE; ++k) {                                        // unconditional (clamped) so all 8 stay in flight
            const int v = k * NTHR + threadIdx.x;
            xv[k] = src4[v < n4 ? v : n4 - 1];
        }
    }
    load_labels<DIR>(ann_j, N, L);

    float cls_scale = 0.f, reg_scale = 0.f, vp_scale = 0.f;
    if (BWD) {
        const ImageStats st = stats[j];
        cls_scale = st.cls_scale; reg_scale = st.reg_scale; vp_scale = st.vp_scale;
    }

    float sums[4] = {0.f, 0.f, 0.f, 0.f};   // focal, smooth-L1, vp, npos
    // ---- phase 1: assignment (+ positives), one lane per anchor, APT anchors per lane.  The 256 consecutive anchors of
    // a wave lie in a small window of the image; their bounding box is reduced across the wave once, and a label
    // whose box misses it has IoU exactly 0 with all 64 (skipping it cannot change the max or the first argmax:
    // IoU >= 0 and ties keep the earlier label).  The test is wave-uniform -- a scalar branch, no divergence -- and
    // removes ~95 % of the IoU evaluations whatever the anchor order (it is merely conservative for odd layouts).
    if (BWD) {                                                // dense dreg: zero the tile, positives overwrite below
        const int64_t nA0 = (A - tile0 < TILE) ? (A - tile0) : TILE;
        float4 *z = reinterpret_cast<float4 *>(dreg + ((int64_t)j * A + tile0) * NREG);
        const int nz = (int)(nA0 * (NREG / 4));
        for (int v = threadIdx.x; v < nz; v += NTHR) z[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    // wave w owns anchors [w*256, w*256+256) of the tile; lane l takes w*256 + u*64 + l in round u (coalesced)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float4 av[APT];
    float lx1 = INFINITY, ly1 = INFINITY, lx2 = -INFINITY, ly2 = -INFINITY;
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        const int64_t ai = tile0 + wv * (64 * APT) + u * 64 + lane;
        av[u] = ai < A ? anchors[ai] : make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
        lx1 = fminf(lx1, av[u].x); ly1 = fminf(ly1, av[u].y);
        lx2 = fmaxf(lx2, av[u].z); ly2 = fmaxf(ly2, av[u].w);
    }
    // one set of four independent reductions per wave: bounding box of its 256 consecutive anchors
    const float wx1 = uniform_f(wave_min_f(lx1)), wy1 = uniform_f(wave_min_f(ly1));
    const float wx2 = uniform_f(wave_max_f(lx2)), wy2 = uniform_f(wave_max_f(ly2));
    float best[APT];
    int arg[APT];
#pragma unroll
    for (int u = 0; u < APT; ++u) { best[u] = 0.f; arg[u] = 0; }
    for (int n = 0; n < L.count; ++n) {
        const float gx1 = uniform_f(L.x1[n]), gy1 = uniform_f(L.y1[n]);
        const float gx2 = uniform_f(L.x2[n]), gy2 = uniform_f(L.y2[n]);
        if (!(wx2 > gx1 && gx2 > wx1 && wy2 > gy1 && gy2 > wy1)) continue;       // every iw or ih <= 0: all IoU = 0
        const float garea = L.area[n];
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const float4 a = av[u];
            float iw = fminf(a.z, gx2) - fmaxf(a.x, gx1);                          // calc_iou's order (D/losses.py:5-22)
            float ih = fminf(a.w, gy2) - fmaxf(a.y, gy1);
            iw = fmaxf(iw, 0.f);
            ih = fmaxf(ih, 0.f);
            const float inter = iw * ih;
            if (inter > 0.f) {                                                     // else 0 / max(ua, 1e-8) == 0
                float ua = ((a.z - a.x) * (a.w - a.y) + garea) - inter;
                ua = fmaxf(ua, 1e-8f);
                const float iou = inter / ua;
                if (iou > best[u]) { best[u] = iou; arg[u] = n; }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        const int al = wv * (64 * APT) + u * 64 + lane;
        const int64_t ai = tile0 + al;
        int state = -1;
        if (ai < A) {
            if (L.count == 0) {
                state = 0;                                                    // empty image: all negative (D/losses.py:58-87)
            } else {
                if (best[u] < 0.4f) state = 0;                                // D/losses.py:121
                if (best[u] >= 0.5f) {                                        // D/losses.py:124
                    const int row = L.row[arg[u]];
                    state = 1 + (int)ann_j[(int64_t)row * COLS + CLS_COL];    // .long() truncation, D/losses.py:131
                    s_pos[atomicAdd(&s_npos, 1)] = al | (row << 16);          // ~0.07 % of anchors: handled below, once
                }
            }
        }
        s_state[al] = state;
    }
    __syncthreads();
    // positives: regression / VP terms (forward) or their gradient rows (backward), one queued anchor per lane.
    // Queue order varies run to run, so the fp32 partial sums of these few terms may differ in the last bit.
    for (int q = threadIdx.x; q < s_npos; q += NTHR) {
        const int e = s_pos[q];
        const int al = e & 0xffff, row = e >> 16;
        const int64_t ai = tile0 + al;
        const float *g = ann_j + (int64_t)row * COLS;
        const float *r = reg + ((int64_t)j * A + ai) * NREG;
        float *dr = BWD ? dreg + ((int64_t)j * A + ai) * NREG : nullptr;
        sums[3] += 1.f;
        if (DIR) positive_dir<BWD>(anchors[ai], g, r, sums[1], sums[2], reg_scale, vp_scale, dr);
        else positive_2d<BWD>(anchors[ai], g, r, sums[1], reg_scale, dr);
    }

    // ---- phase 2: stream the tile's classification values
    float *dst = BWD ? dcls + base : nullptr;
    if ((C & 3) == 0) {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        float4 *dst4 = reinterpret_cast<float4 *>(dst);
        const int n4 = (int)(elems >> 2);
        const int cq = C >> 2;                                                // float4 per anchor
        const int cq_shift = cq == 1 ? 0 : (cq == 2 ? 1 : -1);                // C = 4 / 8: shifts instead of a division
        auto one = [&](int v, const float4 x) {
            const int al = cq_shift >= 0 ? (v >> cq_shift) : v / cq;
            const int c0 = (v - al * cq) << 2;
            const int st = s_state[al];
            const int tc = st - 1 - c0;                                       // lane-local index of the positive class, if any
            // per-anchor weights: negatives 0.75, the positive class 0.25, nothing for an ignored anchor (st < 0)
            const float wn = st < 0 ? 0.f : (BWD ? -0.75f * cls_scale : 0.75f * RN_LN2);
            const float wp = BWD ? 0.25f * cls_scale : 0.25f * RN_LN2;
            const float xs[4] = {x.x, x.y, x.z, x.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool pos = k == tc;                                     // st <= 0 makes tc negative
                o[k] = focal_elem<BWD>(xs[k], pos, pos ? wp : wn);
            }
            if (BWD) dst4[v] = make_float4(o[0], o[1], o[2], o[3]);
            else sums[0] += (o[0] + o[1]) + (o[2] + o[3]);
        };
        if (pre) {
#pragma unroll
            for (int k = 0; k < PRE; ++k) {
                const int v = k * NTHR + threadIdx.x;
                if (v < n4) one(v, xv[k]);
            }
        } else {
#pragma unroll 4
            for (int v = threadIdx.x; v < n4; v += NTHR) one(v, src4[v]);
        }
    } else {
        for (int e = threadIdx.x; e < (int)elems; e += NTHR) {
            const int al = e / C;
            const int c = e - al * C;
            const int st = s_state[al];
            const bool pos = st > 0 && c == st - 1;
            const float wn = st < 0 ? 0.f : (BWD ? -0.75f * cls_scale : 0.75f * RN_LN2);
            const float wp = BWD ? 0.25f * cls_scale : 0.25f * RN_LN2;
            const float o = focal_elem<BWD>(src[e], pos, pos ? wp : wn);
            if (BWD) dst[e] = o;
            else sums[0] += o;
        }
    }
    if (!BWD) {
        block_sum4<NWAVES>(sums, s_red);
        if (threadIdx.x == 0) {
            float4 *p = reinterpret_cast<float4 *>(partials) + ((int64_t)j * gridDim.x + blockIdx.x);
            *p = make_float4(sums[0], sums[1], sums[2], sums[3]);
        }
    }
}

// One workgroup, one wave per image (images beyond 16 loop): the wave adds its image's tile partials in fp64 in a
// fixed lane/tile order (bit-reproducible), checks whether the image has any label row, and derives the per-image
// loss terms and gradient scales; thread 0 then forms the batch means (D/losses.py:152, 350, 304, 359-362).
template <bool DIR>
__global__ __launch_bounds__(1024) void focal_finalize(const float4 *__restrict__ partials, int tiles, int B,
                                                       const float *__restrict__ ann, int N,
                                                       ImageStats *__restrict__ stats, float *__restrict__ losses) {
    constexpr int COLS = DIR ? 27 : 5;
    constexpr int CLS_COL = DIR ? 20 : 4;
    __shared__ double s_l[3][1024];          // per-image loss terms (B <= 1024)
    __shared__ float s_has[1024];
    __shared__ double s_part[16][4][64];     // per wave: lane partials, summed in lane order by 4 lanes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < B; j += 16) {
        double v[4] = {0, 0, 0, 0};
        for (int t = lane; t < tiles; t += 64) {
            const float4 p = partials[(int64_t)j * tiles + t];
            v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s_part[wave][i][lane] = v[i];
        bool any = false;
        for (int r = lane; r < N; r += 64) any |= ann[((int64_t)j * N + r) * COLS + CLS_COL] != -1.0f;
        const bool has = __ballot(any) != 0ull;
        if (lane < 4) {                                      // same-wave LDS traffic: program order suffices
            double t = 0.0;
            for (int k = 0; k < 64; ++k) t += s_part[wave][lane][k];
            s_part[wave][lane][0] = t;
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = s_part[wave][i][0];
            const double npos = v[3];
            ImageStats st;
            st.npos = (float)npos;
            st.has_labels = has ? 1.f : 0.f;
            st.pad[0] = st.pad[1] = st.pad[2] = 0.f;
            const double nvals = DIR ? 20.0 : 4.0;
            double lc, lr = 0.0, lv = 0.0;
            if (!has) {
                lc = v[0];                                                    // raw sum, not normalised (D/losses.py:58-87)
                st.cls_scale = 1.0f; st.reg_scale = 0.f; st.vp_scale = 0.f;
            } else {
                const double dn = npos > 1.0 ? npos : 1.0;                    // clamp(min=1), D/losses.py:152
                lc = v[0] / dn;
                st.cls_scale = (float)(1.0 / dn);
                if (npos > 0.0) {
                    lr = v[1] / (npos * nvals);                               // mean over [P,20] / [P,4]
                    lv = v[2] / (3.0 * npos);                                 // mean over P of (sum_k)/3
                    st.reg_scale = (float)(1.0 / (npos * nvals));
                    st.vp_scale = (float)(1.0 / (3.0 * npos));
                } else {
                    st.reg_scale = 0.f; st.vp_scale = 0.f;
                }
            }
            s_l[0][j] = lc; s_l[1][j] = lr; s_l[2][j] = lv;
            s_has[j] = st.has_labels;
            stats[j] = st;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum[3] = {0, 0, 0};
        int vp_images = 0;
        for (int j = 0; j < B; ++j) {
            sum[0] += s_l[0][j]; sum[1] += s_l[1][j]; sum[2] += s_l[2][j];
            vp_images += s_has[j] != 0.f;
        }
        losses[0] = (float)(sum[0] / B);
        losses[1] = (float)(sum[1] / B);
        // vp: mean over images that have labels; none -> 0/0 = NaN (the reference raises, D/losses.py:362)
        losses[2] = DIR ? (float)(sum[2] / (double)vp_images) : 0.f;
        for (int j = 0; j < B; ++j) {                                          // fold the batch means into the scales
            stats[j].cls_scale /= (float)B;
            stats[j].reg_scale /= (float)B;
            stats[j].vp_scale = vp_images > 0 ? stats[j].vp_scale / (float)vp_images : 0.f;
        }
    }
}

// Multiply the per-image scales by the incoming loss gradients (device scalars: no host sync).
__global__ void scale_kernel(const ImageStats *__restrict__ in, const float *__restrict__ g, int B,
                             ImageStats *__restrict__ out) {
    const int j = threadIdx.x;
    if (j < B) {
        ImageStats s = in[j];
        s.cls_scale *= g[0]; s.reg_scale *= g[1]; s.vp_scale *= g[2];
        out[j] = s;
    }
}

static int check_args(int B, int64_t A, int C, int N) {
    if (B <= 0 || A <= 0 || C <= 0 || N < 0) return RN_EINVAL;
    if (N > RN_MAX_GT) return RN_ETOOMANY;
    if (B > 1024) return RN_EINVAL;
    if ((A + TILE - 1) / TILE > 0x7fffffffLL || B > 65535) return RN_EINVAL;
    return RN_OK;
}

extern "C" int rn_focal_loss_fwd(const float *cls, const float *reg, const float *anchors, const float *ann, int B,
                                 int64_t A, int C, int N, int directional, void *workspace, float *losses,
                                 void *stream) {
    const int rc = check_args(B, A, C, N);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (int)((A + TILE - 1) / TILE);
    ImageStats *stats = reinterpret_cast<ImageStats *>(workspace);
    float *partials = reinterpret_cast<float *>(stats + 2 * B);
    const dim3 grid(tiles, B), block(NTHR);
    const float4 *anc = reinterpret_cast<const float4 *>(anchors);
    if (directional) {
        hipLaunchKernelGGL((focal_kernel<true, false>), grid, block, 0, s, cls, reg, anc, ann, A, C, N, partials,
                           (const ImageStats *)nullptr, (float *)nullptr, (float *)nullptr);
        hipLaunchKernelGGL(focal_finalize<true>, dim3(1), dim3(1024), 0, s, (const float4 *)partials, tiles, B, ann, N,
                           stats, losses);
    } else {
        hipLaunchKernelGGL((focal_kernel<false, false>), grid, block, 0, s, cls, reg, anc, ann, A, C, N, partials,
                           (const ImageStats *)nullptr, (float *)nullptr, (float *)nullptr);
        hipLaunchKernelGGL(focal_finalize<false>, dim3(1), dim3(1024), 0, s, (const float4 *)partials, tiles, B, ann, N,
                           stats, losses);
    }
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_focal_loss_bwd(const float *cls, const float *reg, const float *anchors, const float *ann, int B,
                                 int64_t A, int C, int N, int directional, const void *workspace,
                                 const float *grad_losses, float *dcls, float *dreg, void *stream) {
    const int rc = check_args(B, A, C, N);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (int)((A + TILE - 1) / TILE);
    const ImageStats *stats = reinterpret_cast<const ImageStats *>(workspace);
    ImageStats *scaled = const_cast<ImageStats *>(stats) + B;                  // second block of the workspace
    hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(1024), 0, s, stats, grad_losses, B, scaled);
    const dim3 grid(tiles, B), block(NTHR);
    const float4 *anc = reinterpret_cast<const float4 *>(anchors);
    if (directional)
        hipLaunchKernelGGL((focal_kernel<true, true>), grid, block, 0, s, cls, reg, anc, ann, A, C, N, (float *)nullptr,
                           (const ImageStats *)scaled, dcls, dreg);
    else
        hipLaunchKernelGGL((focal_kernel<false, true>), grid, block, 0, s, cls, reg, anc, ann, A, C, N, (float *)nullptr,
                           (const ImageStats *)scaled, dcls, dreg);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ----------------------------------------------------------------------------------------------------------
template <bool DIR>
__global__ __launch_bounds__(NTHR) void assign_kernel(const float4 *__restrict__ anchors, const float *__restrict__ ann,
                                                      int64_t A, int N, float *__restrict__ iou_max,
                                                      int32_t *__restrict__ argmax, int32_t *__restrict__ state) {
    constexpr int COLS = DIR ? 27 : 5;
    __shared__ LabelLds L;
    const int j = blockIdx.y;
    load_labels<DIR>(ann + (int64_t)j * N * COLS, N, L);
    const int64_t ai = (int64_t)blockIdx.x * NTHR + threadIdx.x;
    if (ai >= A) return;
    float best = 0.f; int arg = -1; int st = 0;
    if (L.count > 0) {
        match_anchor(anchors[ai], L, best, arg);
        st = -1;
        if (best < 0.4f) st = 0;
        if (best >= 0.5f) st = 1;
    }
    iou_max[(int64_t)j * A + ai] = best;
    argmax[(int64_t)j * A + ai] = arg;
    state[(int64_t)j * A + ai] = st;
}

extern "C" int rn_assign(const float *anchors, const float *ann, int B, int64_t A, int N, int directional,
                         float *iou_max, int32_t *argmax, int32_t *state, void *stream) {
    const int rc = check_args(B, A, 1, N);
    if (rc) return rc;
    const dim3 grid((unsigned)((A + NTHR - 1) / NTHR), B), block(NTHR);
    const float4 *anc = reinterpret_cast<const float4 *>(anchors);
    if (directional)
        hipLaunchKernelGGL(assign_kernel<true>, grid, block, 0, (hipStream_t)stream, anc, ann, A, N, iou_max, argmax, state);
    else
        hipLaunchKernelGGL(assign_kernel<false>, grid, block, 0, (hipStream_t)stream, anc, ann, A, N, iou_max, argmax, state);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// calc_iou as a standalone op (D/losses.py:5-22): one lane per (anchor, label) pair, label index fastest.
__global__ void pairwise_iou_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, float *__restrict__ out,
                                    int64_t A, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A * N) return;
    const int64_t ai = i / N;
    const int n = (int)(i - ai * N);
    const float4 p = a[ai], q = b[n];
    const float area_b = (q.z - q.x) * (q.w - q.y);
    float iw = fminf(p.z, q.z) - fmaxf(p.x, q.x);
    float ih = fminf(p.w, q.w) - fmaxf(p.y, q.y);
    iw = fmaxf(iw, 0.f);
    ih = fmaxf(ih, 0.f);
    float ua = ((p.z - p.x) * (p.w - p.y) + area_b) - iw * ih;
    ua = fmaxf(ua, 1e-8f);
    out[i] = (iw * ih) / ua;
}

extern "C" int rn_pairwise_iou(const float *a, const float *b, float *iou, int64_t A, int N, void *stream) {
    if (A <= 0 || N <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(pairwise_iou_kernel, dim3(rn_blocks(A * N, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b), iou, A, N);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
