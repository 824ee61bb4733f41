// Fused anchor<->label IoU matching + focal classification loss + smooth-L1 box loss + vanishing-point loss.
//
// Replaces FocalLoss.forward of the reference -- directional 3D variant D/losses.py:27-362 and 2D variant
// R/losses.py:27-177 -- together with calc_iou (D/losses.py:5-22).  The reference walks the batch in a Python
// loop and runs ~40 small torch kernels per image, materialising an [A,N] IoU matrix, an [A,C] target matrix
// and several [A,C] temporaries (348 ms forward at B=8 on 8 CPU threads).
//
// One launch covers the whole batch, forward (including the batch reduction) and backward alike.  The grid is
// persistent: B*G workgroups, all resident; workgroup g of image j stages the image's label rows in LDS once and each
// of its waves walks 128-anchor units, software-pipelined (the next unit's anchors and classification values are
// requested before the current unit is processed).  Per unit:
//   assignment  one lane per anchor; the wave reduces the bounding box of its 128 consecutive anchors (DPP), lane r
//               tests label r against it, a ballot leaves the labels that can overlap at all (the others have IoU
//               exactly 0); those get the full IoU -- fp32, one rounding per operation in the reference's order (this
//               file is compiled with -ffp-contract=off, so the 0.4 / 0.5 band comparisons and the first-maximum
//               argmax are bit-identical to torch CPU).  For the directional variant the label box is the min/max
//               envelope of the 8 corners (D/losses.py:93-107), NOT label cols 16:20;
//   stream      the unit's 128*C classification values as float4 (16 B per lane, consecutive lanes consecutive
//               addresses); ~95 % of the units hold only negative anchors and take a packed-fp32 path without any
//               per-value select; forward accumulates the focal sum, backward writes dcls;
//   positives   (~0.07 % of anchors) are queued per wave in anchor order and evaluated when no prefetch is in flight.
// Sums leave a workgroup as one 4-float partial, published with agent-scope atomics; the last workgroup to finish
// adds them per image in fp64 in a fixed order and forms the batch means.  No float atomics, no order-dependent sums:
// the losses are bit-reproducible run to run.
//
// Roofline: HBM.  Forward algorithmic bytes = B*A*C*4 (cls) + A*16 (anchors; read once per XCD, L2-resident for the other
// images) + B*N*cols*4 = 105.9 MB at B=8, A=389 205, C=8; measured HBM reads 104 MB.  Backward adds
// the dcls and dreg writes (99.6 + 149.5 MB).  What bounds it in practice (profiles/r01_loss_analysis.txt,
// r02_loss_analysis.txt): a bare 100 MB streaming read in the same launch shape takes 17 us; of the forward's 31 us the
// loads-only skeleton of this kernel is 21.8, the focal arithmetic adds 1.6, the assignment 3.0, publishing the partials
// 0.6 and the two-level completion chain 4.3 (five dependent device-scope round trips of ~0.8 us).
#include <math.h>
#include <stdlib.h>

#include "common.h"

#ifndef FOCAL_NTHR
#define FOCAL_NTHR 256
#endif
#define NTHR FOCAL_NTHR   // threads per workgroup
#define APT 4             // anchors per thread of rn_assign
#define TILE (NTHR * APT) // rn_assign: 1024 anchors per workgroup
#define NWAVES (NTHR / 64)
#define UAPT 2            // fused loss: anchors per lane per unit
#define UNIT (64 * UAPT)  // fused loss: a wave's work item = 128 consecutive anchors of one image
#define QCAP 256          // fused loss: capacity of a wave's queue of positive anchors (>= 2 * UNIT)
#ifndef RN_FOCAL_WAVES_PER_SIMD
#define RN_FOCAL_WAVES_PER_SIMD 4
#endif
#define RN_FOCAL_WAVES_NOTE 4  // (HIP: second launch-bound = waves per SIMD) the register allocator must leave room for (128 VGPRs: no spills)
#define RN_FOCAL_RESIDENT 768   // workgroups of the persistent grid (3 per CU).  Measured at B=8, A=389 205: 512..768 is
                                // the optimum; more workgroups lengthen the same-address completion-counter queue
                                // (~8 ns each) faster than they add memory-level parallelism

// Workgroups per image of the persistent loss kernel: all B*G workgroups are resident in one round; G is a multiple
// of 8 so that workgroup g of every image lands on XCD g%8 (ids are dealt round-robin) and the images share the
// anchors of "their" units in that XCD's L2.
static inline int focal_resident() {                             // RN_FOCAL_RESIDENT in the environment: tuning sweeps only
    static const int v = [] { const char *e = getenv("RN_FOCAL_RESIDENT"); const int x = e ? atoi(e) : 0; return x >= 8 ? x : RN_FOCAL_RESIDENT; }();
    return v;
}
static inline int focal_groups(int B, int64_t A) {
    const int64_t units = (A + UNIT - 1) / UNIT;
    int64_t G = (focal_resident() / B) & ~7;
    if (G < 8) G = 8;
    const int64_t need = (units + NWAVES - 1) / NWAVES;          // more workgroups than units/4 would idle
    if (G > need) G = need >= 8 ? ((need + 7) & ~7LL) : need;
    return (int)(G < 1 ? 1 : G);
}

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

// Workspace layout: [64 B] batch completion counter | [B] per-image completion counters (padded to 64 B) -- these first
//                   rn_focal_workspace_zero_bytes(B) bytes must be ZERO on entry to rn_focal_loss_fwd; left zero on exit --
//                   | [B] ImageStats | [B][8] double per-image loss terms | [B][G] float4 partial sums
#define RN_WS_HEAD 64
#define RN_TERMS 8        // doubles per image: cls, reg, vp loss terms, has_labels, npos, (3 unused)
static inline int64_t focal_counter_bytes(int B) { return ((int64_t)B * 4 + 63) / 64 * 64; }
extern "C" int64_t rn_focal_workspace_zero_bytes(int B) { return B > 0 ? RN_WS_HEAD + focal_counter_bytes(B) : 0; }
extern "C" int64_t rn_focal_workspace_bytes(int B, int64_t A) {
    if (B <= 0 || A <= 0) return 0;
    return RN_WS_HEAD + focal_counter_bytes(B) + (int64_t)B * sizeof(ImageStats) + (int64_t)B * RN_TERMS * sizeof(double) +
           (int64_t)B * focal_groups(B, A) * 4 * sizeof(float);
}

// ----------------------------------------------------------------------------------------------------------
struct LabelLds {
    float x1[RN_MAX_GT], y1[RN_MAX_GT], x2[RN_MAX_GT], y2[RN_MAX_GT], area[RN_MAX_GT];
    int row[RN_MAX_GT];     // original row index in ann[j]
    int count;
};

// One label row reduced to its matching box: valid iff the class column != -1 (D/losses.py:54, R/losses.py:47);
// the directional box is the min/max envelope of the 8 corners (D/losses.py:93-107).
template <bool DIR>
__device__ __forceinline__ void load_label_row(const float *__restrict__ ann_j, int N, int r, float &bx1, float &by1,
                                               float &bx2, float &by2, bool &valid) {
    constexpr int COLS = DIR ? 27 : 5;
    constexpr int CLS_COL = DIR ? 20 : 4;
    valid = false;
    bx1 = by1 = bx2 = by2 = 0.f;
    if (r < N) {
        const float *p = ann_j + (int64_t)r * COLS;
        valid = p[CLS_COL] != -1.0f;
        if (DIR) {
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
}

// The same from a raw row staged in LDS (row stride COLS words is odd: conflict-free across lanes).
template <bool DIR>
__device__ __forceinline__ void label_row_from_lds(const float *__restrict__ raw, int n_rows, int r, float &bx1, float &by1,
                                                   float &bx2, float &by2, bool &valid) {
    constexpr int COLS = DIR ? 27 : 5;
    constexpr int CLS_COL = DIR ? 20 : 4;
    valid = false;
    bx1 = by1 = bx2 = by2 = 0.f;
    if (r < n_rows) {
        const float *p = raw + r * COLS;
        valid = p[CLS_COL] != -1.0f;
        if (DIR) {
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
}

// Collect valid rows in their original order into LDS (rn_assign): wave 0 scans rows in chunks of 64, ordered
// compaction by ballot.
template <bool DIR>
__device__ __forceinline__ void load_labels(const float *__restrict__ ann_j, int N, LabelLds &L) {
    if (threadIdx.x < 64) {
        int base = 0;
        for (int r0 = 0; r0 < N; r0 += 64) {
            const int r = r0 + threadIdx.x;
            bool valid;
            float bx1, by1, bx2, by2;
            load_label_row<DIR>(ann_j, N, r, bx1, by1, bx2, by2, valid);
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
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

// Epilogue of the forward, two levels so that nothing waits in one long same-address queue and the per-image work runs
// in parallel (profiles/r01_loss_analysis.txt: 768 arrivals at one counter ~6 us, then a serial epilogue ~7 us):
//   level 1  the LAST workgroup of image j to finish (per-image counter: G arrivals) adds that image's G partials -- all
//            256 threads, fp64, a fixed strided + butterfly order (bit-reproducible) --, checks whether the image has any
//            label row, and publishes the per-image loss terms;
//   level 2  the last of those B workgroups (batch counter: B arrivals) forms the batch means and writes the per-image
//            gradient scales the backward kernel reads (D/losses.py:152, 350, 304, 359-362).
__device__ __forceinline__ double shfl_xor_f64(double v, int off) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off, 64);
    hi = __shfl_xor(hi, off, 64);
    return __hiloint2double(hi, lo);
}
// Sum of v[0..NV) over the 256 threads of the workgroup in a fixed order; valid in thread 0.  red: 4*NV doubles of LDS.
template <int NV>
__device__ __forceinline__ void block_sum_f64(double (&v)[NV], double *red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += shfl_xor_f64(v[i], off);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[w * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double t = 0.0;
            for (int k = 0; k < NWAVES; ++k) t += red[k * NV + i];            // fixed order
            v[i] = t;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void publish_f64(double *p, double v) {     // agent scope, no L2 writeback (see the partials)
    (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double fetch_f64(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}

template <bool DIR>
__device__ __forceinline__ void focal_finalize_image(const float4 *__restrict__ partials, int G, int j,
                                                     const float *__restrict__ ann, int N, double *__restrict__ terms,
                                                     double *s_red) {
    constexpr int COLS = DIR ? 27 : 5;
    constexpr int CLS_COL = DIR ? 20 : 4;
    double v[4] = {0, 0, 0, 0};
    for (int t0 = 0; t0 < G; t0 += NTHR * 2) {                                // 2 independent partials in flight per thread
        float4 p[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int t = t0 + k * NTHR + threadIdx.x;
            const unsigned long long *q = reinterpret_cast<const unsigned long long *>(partials) +
                                          2 * ((int64_t)j * G + (t < G ? t : G - 1));
            const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p[k] = make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                               __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (t0 + k * NTHR + threadIdx.x < G) { v[0] += p[k].x; v[1] += p[k].y; v[2] += p[k].z; v[3] += p[k].w; }
    }
    bool any = false;
    for (int r = threadIdx.x; r < N; r += NTHR) any |= ann[((int64_t)j * N + r) * COLS + CLS_COL] != -1.0f;
    const bool has = __syncthreads_or(any) != 0;
    block_sum_f64<4>(v, s_red);
    if (threadIdx.x == 0) {
        const double npos = v[3];
        const double nvals = DIR ? 20.0 : 4.0;
        double lc, lr = 0.0, lv = 0.0;
        if (!has) {
            lc = v[0];                                                        // raw sum, not normalised (D/losses.py:58-87)
        } else {
            const double dn = npos > 1.0 ? npos : 1.0;                        // clamp(min=1), D/losses.py:152
            lc = v[0] / dn;
            if (npos > 0.0) {
                lr = v[1] / (npos * nvals);                                   // mean over [P,20] / [P,4]
                lv = v[2] / (3.0 * npos);                                     // mean over P of (sum_k)/3
            }
        }
        double *t = terms + (int64_t)j * RN_TERMS;
        publish_f64(t + 0, lc); publish_f64(t + 1, lr); publish_f64(t + 2, lv);
        publish_f64(t + 3, has ? 1.0 : 0.0); publish_f64(t + 4, npos);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <bool DIR>
__device__ __forceinline__ void focal_finalize_batch(int B, const double *__restrict__ terms, ImageStats *__restrict__ stats,
                                                     float *__restrict__ losses, double *s_red, double *s_bc) {
    double v[4] = {0, 0, 0, 0};                                               // cls, reg, vp sums, images with labels
    for (int j = threadIdx.x; j < B; j += NTHR) {
        const double *t = terms + (int64_t)j * RN_TERMS;
        v[0] += fetch_f64(t + 0); v[1] += fetch_f64(t + 1); v[2] += fetch_f64(t + 2);
        v[3] += fetch_f64(t + 3) != 0.0 ? 1.0 : 0.0;
    }
    block_sum_f64<4>(v, s_red);
    if (threadIdx.x == 0) {
        losses[0] = (float)(v[0] / B);
        losses[1] = (float)(v[1] / B);
        // vp: mean over images that have labels; none -> 0/0 = NaN (the reference raises, D/losses.py:362)
        losses[2] = DIR ? (float)(v[2] / v[3]) : 0.f;
        s_bc[0] = v[3];
    }
    __syncthreads();
    const int vp_images = (int)s_bc[0];
    const double nvals = DIR ? 20.0 : 4.0;
    for (int j = threadIdx.x; j < B; j += NTHR) {                              // per-image scales with the batch means folded in
        const double *t = terms + (int64_t)j * RN_TERMS;
        const bool has = fetch_f64(t + 3) != 0.0;
        const double npos = fetch_f64(t + 4);
        ImageStats st;
        st.npos = (float)npos;
        st.has_labels = has ? 1.f : 0.f;
        st.pad[0] = st.pad[1] = st.pad[2] = 0.f;
        st.cls_scale = 1.0f; st.reg_scale = 0.f; st.vp_scale = 0.f;
        if (has) {
            const double dn = npos > 1.0 ? npos : 1.0;
            st.cls_scale = (float)(1.0 / dn);
            if (npos > 0.0) {
                st.reg_scale = (float)(1.0 / (npos * nvals));
                st.vp_scale = (float)(1.0 / (3.0 * npos));
            }
        }
        st.cls_scale = st.cls_scale / (float)B;
        st.reg_scale = st.reg_scale / (float)B;
        st.vp_scale = vp_images > 0 ? st.vp_scale / (float)vp_images : 0.f;
        stats[j] = st;
    }
}

// ----------------------------------------------------------------------------------------------------------
// Persistent, software-pipelined fused loss.  Grid = B*G workgroups, all resident at once; workgroup g of image j
// stages the image's label rows in LDS once, then each of its 4 waves walks units u = g*4 + w, += G*4 (a unit =
// 128 consecutive anchors).  Per unit a wave
//   (0) has already requested the NEXT unit's anchors and classification values (register double buffer), so HBM
//       latency and the VALU work of the current unit overlap inside every wave;
//   (1) assigns its 2 anchors per lane (window test by ballot, exact IoU for the few labels that can overlap);
//   (2) hands states to the streaming lanes through a wave-private LDS row (no workgroup barrier in the loop) and
//       streams the unit's classification values;
//   (3) evaluates its positives, compacted in (anchor) order by ballot prefix -- deterministic, so the whole loss is
//       bit-reproducible run to run.
// CQ = C/4 when C is 4 or 8 (values prefetched into registers), 0 for any other C (streamed without prefetch).
// A template parameter, not a runtime test: loads inside conditional blocks make the compiler's s_waitcnt
// accounting pessimistic.
template <bool DIR, bool BWD, int CQ>
__global__ __launch_bounds__(NTHR, RN_FOCAL_WAVES_PER_SIMD) void focal_kernel(const float *__restrict__ cls, const float *__restrict__ reg,
                                                     const float4 *__restrict__ anchors, const float *__restrict__ ann,
                                                     int64_t A, int C, int N, float *__restrict__ partials,
                                                     ImageStats *__restrict__ stats, float *__restrict__ dcls,
                                                     float *__restrict__ dreg, int G, int units,
                                                     unsigned *__restrict__ counter, double *__restrict__ terms,
                                                     float *__restrict__ losses, const float *__restrict__ grad_losses) {
    constexpr int COLS = DIR ? 27 : 5;
    constexpr int CLS_COL = DIR ? 20 : 4;
    constexpr int NREG = DIR ? 12 : 4;
    constexpr int NX = CQ > 0 ? UAPT * CQ : 1;               // float4 per lane per unit held in registers
    extern __shared__ float s_raw[];                         // the image's label rows, raw: N*COLS floats (dynamic)
    __shared__ int s_wstate[NWAVES][UNIT];                   // per wave: -1 ignore, 0 negative, 1 + class for a positive
    __shared__ int s_q[2][NWAVES][QCAP];                     // per wave: queue of positives -- [0] anchor index, [1] label row
    __shared__ float s_red[NWAVES * 4];
    __shared__ int s_last;
    static_assert(sizeof(int) * 2 * NWAVES * QCAP >= sizeof(double) * (NWAVES * 4 + 1), "epilogue scratch reuses the queues");

    const int j = blockIdx.x / G, g = blockIdx.x - j * G;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // wave-uniform: unit ids and bases stay scalar
    const float *ann_j = ann + (int64_t)j * N * COLS;
    float cls_scale = 0.f, reg_scale = 0.f, vp_scale = 0.f;
    if (BWD) {                                              // per-image scales x incoming loss gradients (device scalars)
        const ImageStats st = stats[j];
        cls_scale = st.cls_scale * grad_losses[0];
        reg_scale = st.reg_scale * grad_losses[1];
        vp_scale = st.vp_scale * grad_losses[2];
    }
    float sums[4] = {0.f, 0.f, 0.f, 0.f};   // focal, smooth-L1, vp, npos
    f32x2 neg2 = {0.f, 0.f};                // fast path: sum of p^2 * log2(1-p) over plain negatives

    // ---- (0) request a unit: anchors first, classification values behind them (vmcnt retires in order)
    auto issue = [&](int u, float4 (&av)[UAPT], float4 (&xv)[NX]) {
        const int64_t a0 = (int64_t)u * UNIT;                                  // scalar: u is wave-uniform
        const unsigned nA = (unsigned)(A - a0 < UNIT ? A - a0 : UNIT);
        const float4 *ab = anchors + a0;
#pragma unroll
        for (int uu = 0; uu < UAPT; ++uu) {
            const unsigned o = uu * 64 + lane;
            av[uu] = ab[o < nA ? o : nA - 1];
        }
        if (CQ > 0) {
            const float4 *src4 = reinterpret_cast<const float4 *>(cls + ((int64_t)j * A + a0) * C);
            const unsigned n4 = nA * CQ;
#pragma unroll
            for (int k = 0; k < NX; ++k) {                                     // unconditional (clamped): all stay in flight
                const unsigned v = k * 64 + lane;
                xv[k] = src4[v < n4 ? v : n4 - 1];
            }
        }
    };

    // the first unit is requested before anything else: its latency overlaps the label staging
    const int stride = G * NWAVES;
    int u = g * NWAVES + wv;
    float4 avA[UAPT], avB[UAPT], xvA[NX], xvB[NX];                             // register double buffer (ping-pong: no copies)
    if (u < units) issue(u, avA, xvA);

    // labels: N*COLS contiguous floats, fetched coalesced once per workgroup (per-lane row gathers from global would
    // cost more L2 requests than the classification stream itself)
    for (int i = threadIdx.x; i < N * COLS; i += NTHR) s_raw[i] = ann_j[i];
    __syncthreads();
    // rows 0..63 live in registers for the whole kernel, one row per lane; rows beyond 64 are re-read from LDS per unit
    const int n0 = N < 64 ? N : 64;
    float bx1, by1, bx2, by2;
    bool bvalid;
    label_row_from_lds<DIR>(s_raw, n0, lane, bx1, by1, bx2, by2, bvalid);
    const float bcls = lane < n0 ? s_raw[lane * COLS + CLS_COL] : 0.f;
    const float barea = (bx2 - bx1) * (by2 - by1);                             // D/losses.py:6
    bool any_valid = __ballot(bvalid) != 0ull;
    for (int c0 = 64; c0 < N; c0 += 64)
        any_valid |= __ballot(c0 + lane < N && s_raw[(c0 + lane) * COLS + CLS_COL] != -1.0f) != 0ull;


    // one label against the wave's anchors; rows arrive in increasing order, so a strict > keeps the first maximum
    auto match_one = [&](const float4 (&av)[UAPT], float (&best)[UAPT], int (&arg)[UAPT], float (&cbest)[UAPT], int row,
                         float gx1, float gy1, float gx2, float gy2, float garea, float gcls) {
#pragma unroll
        for (int uu = 0; uu < UAPT; ++uu) {
            const float4 a = av[uu];
            float iw = fminf(a.z, gx2) - fmaxf(a.x, gx1);                          // calc_iou's order (D/losses.py:5-22)
            float ih = fminf(a.w, gy2) - fmaxf(a.y, gy1);
            iw = fmaxf(iw, 0.f);
            ih = fmaxf(ih, 0.f);
            const float inter = iw * ih;
            if (inter > 0.f) {                                                     // else 0 / max(ua, 1e-8) == 0
                float ua = ((a.z - a.x) * (a.w - a.y) + garea) - inter;
                ua = fmaxf(ua, 1e-8f);
                const float iou = inter / ua;
                if (iou > best[uu]) { best[uu] = iou; arg[uu] = row; cbest[uu] = gcls; }
            }
        }
    };

    int qn = 0;                                                                // queued positives of this wave (uniform)
    auto process = [&](int u, float4 (&av)[UAPT], const float4 (&xv)[NX]) {
        const int64_t a0 = (int64_t)u * UNIT;
        const int nA = (int)(A - a0 < UNIT ? A - a0 : UNIT);
        // ---- (1) assignment.  The wave's 128 consecutive anchors lie in a small window of the image; its bounding box
        // is reduced once (DPP), lane r tests label r against it, and one ballot yields the labels that can overlap any
        // of them.  A label outside the window has IoU exactly 0 with all of them (skipping it cannot change the max or
        // the first argmax: IoU >= 0 and ties keep the earlier row).
        float lx1 = INFINITY, ly1 = INFINITY, lx2 = -INFINITY, ly2 = -INFINITY;
#pragma unroll
        for (int uu = 0; uu < UAPT; ++uu) {
            if (uu * 64 + lane >= nA) av[uu] = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);   // neutral, IoU 0
            lx1 = fminf(lx1, av[uu].x); ly1 = fminf(ly1, av[uu].y);
            lx2 = fmaxf(lx2, av[uu].z); ly2 = fmaxf(ly2, av[uu].w);
        }
        wave_bbox_dpp(lx1, ly1, lx2, ly2);
        const float wx1 = lx1, wy1 = ly1, wx2 = lx2, wy2 = ly2;
        float best[UAPT], cbest[UAPT];
        int arg[UAPT];
#pragma unroll
        for (int uu = 0; uu < UAPT; ++uu) { best[uu] = 0.f; arg[uu] = 0; cbest[uu] = 0.f; }
        unsigned long long m = __ballot(bvalid && wx2 > bx1 && bx2 > wx1 && wy2 > by1 && by2 > wy1);
        while (m) {                                                            // wave-uniform: scalar loop over set bits
            const int n = __builtin_ctzll(m);
            m &= m - 1;
            match_one(av, best, arg, cbest, n, rn_readlane_f(bx1, n), rn_readlane_f(by1, n), rn_readlane_f(bx2, n),
                      rn_readlane_f(by2, n), rn_readlane_f(barea, n), rn_readlane_f(bcls, n));
        }
        for (int c0 = 64; c0 < N; c0 += 64) {                                  // N > 64: further rows straight from LDS
            const int nrow = N - c0 < 64 ? N - c0 : 64;
            float cx1, cy1, cx2, cy2;
            bool cvalid;
            label_row_from_lds<DIR>(s_raw + c0 * COLS, nrow, lane, cx1, cy1, cx2, cy2, cvalid);
            const float ccls = lane < nrow ? s_raw[(c0 + lane) * COLS + CLS_COL] : 0.f;
            const float carea = (cx2 - cx1) * (cy2 - cy1);
            unsigned long long mc = __ballot(cvalid && wx2 > cx1 && cx2 > wx1 && wy2 > cy1 && cy2 > wy1);
            while (mc) {
                const int n = __builtin_ctzll(mc);
                mc &= mc - 1;
                match_one(av, best, arg, cbest, c0 + n, rn_readlane_f(cx1, n), rn_readlane_f(cy1, n), rn_readlane_f(cx2, n),
                          rn_readlane_f(cy2, n), rn_readlane_f(carea, n), rn_readlane_f(ccls, n));
            }
        }
        int state[UAPT];
        bool pos[UAPT];
        bool plain = true;                                                     // lane's anchors are plain negatives (or past the end)
#pragma unroll
        for (int uu = 0; uu < UAPT; ++uu) {
            const int al = uu * 64 + lane;
            state[uu] = -1;
            pos[uu] = false;
            if (al < nA) {
                if (!any_valid) {
                    state[uu] = 0;                                            // empty image: all negative (D/losses.py:58-87)
                } else {
                    if (best[uu] < 0.4f) state[uu] = 0;                       // D/losses.py:121
                    if (best[uu] >= 0.5f) {                                   // D/losses.py:124
                        state[uu] = 1 + (int)cbest[uu];                       // .long() truncation, D/losses.py:131
                        pos[uu] = true;
                    }
                }
                plain = plain && state[uu] == 0;
            }
        }
        const int64_t base = ((int64_t)j * A + a0) * C;
        const float *src = cls + base;
        float *dst = BWD ? dcls + base : nullptr;
        const int n4 = nA * CQ;
        // ---- (2) stream the unit's classification values.  Nearly every unit (~95 %) holds nothing but negative
        // anchors: a = 1-p, m = p for every value -- packed fp32 math, no per-value selects, no state exchange; the
        // forward keeps the common weight 0.75*ln2 out of the sum (applied once per lane at the end).
        if (CQ > 0 && __ballot(!plain) == 0ull) {
            float4 *dst4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                const int v = k * 64 + lane;
                if (v < n4) {
                    const float4 x = xv[k];
                    const f32x2 p01 = {__builtin_amdgcn_fmed3f(x.x, 1e-4f, 1.0f - 1e-4f), __builtin_amdgcn_fmed3f(x.y, 1e-4f, 1.0f - 1e-4f)};
                    const f32x2 p23 = {__builtin_amdgcn_fmed3f(x.z, 1e-4f, 1.0f - 1e-4f), __builtin_amdgcn_fmed3f(x.w, 1e-4f, 1.0f - 1e-4f)};
                    const f32x2 q01 = 1.0f - p01, q23 = 1.0f - p23;
                    const f32x2 l01 = {__builtin_amdgcn_logf(q01.x), __builtin_amdgcn_logf(q01.y)};
                    const f32x2 l23 = {__builtin_amdgcn_logf(q23.x), __builtin_amdgcn_logf(q23.y)};
                    if (!BWD) {
                        neg2 += (p01 * p01) * l01;
                        neg2 += (p23 * p23) * l23;
                    } else {
                        const f32x2 r01 = {__builtin_amdgcn_rcpf(q01.x), __builtin_amdgcn_rcpf(q01.y)};
                        const f32x2 r23 = {__builtin_amdgcn_rcpf(q23.x), __builtin_amdgcn_rcpf(q23.y)};
                        const float wl = -0.75f * cls_scale;
                        const f32x2 g01 = wl * ((2.0f * RN_LN2) * (p01 * l01) - (p01 * p01) * r01);
                        const f32x2 g23 = wl * ((2.0f * RN_LN2) * (p23 * l23) - (p23 * p23) * r23);
                        // p == x exactly when x lies inside the clamp range; outside it the clamp passes no gradient
                        dst4[v] = make_float4(p01.x == x.x ? g01.x : 0.f, p01.y == x.y ? g01.y : 0.f,
                                              p23.x == x.z ? g23.x : 0.f, p23.y == x.w ? g23.y : 0.f);
                    }
                }
            }
        } else {
            // general unit: states go to the streaming lanes through the wave-private LDS row, positives are queued in
            // anchor order (ballot prefix: deterministic)
#pragma unroll
            for (int uu = 0; uu < UAPT; ++uu) {
                const int al = uu * 64 + lane;
                s_wstate[wv][al] = state[uu];
                const unsigned long long pm = __ballot(pos[uu]);               // ~0.07 % of anchors
                if (pos[uu]) {
                    const int q = qn + __popcll(pm & ((1ull << lane) - 1ull));
                    s_q[0][wv][q] = (int)(a0 + al);
                    s_q[1][wv][q] = arg[uu];
                }
                qn += __popcll(pm);
            }
            __builtin_amdgcn_wave_barrier();                                   // wave-private LDS rows: program order suffices
            if (CQ > 0) {
                float4 *dst4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
                for (int k = 0; k < NX; ++k) {
                    const int v = k * 64 + lane;
                    if (v < n4) {
                        const float4 x = xv[k];
                        const int al = CQ == 2 ? (v >> 1) : v;
                        const int c0 = CQ == 2 ? ((v & 1) << 2) : 0;
                        const int st = s_wstate[wv][al];
                        const int tc = st - 1 - c0;                           // lane-local index of the positive class, if any
                        // per-anchor weights: negatives 0.75, the positive class 0.25, nothing for an ignored anchor
                        const float wn = st < 0 ? 0.f : (BWD ? -0.75f * cls_scale : 0.75f * RN_LN2);
                        const float wp = BWD ? 0.25f * cls_scale : 0.25f * RN_LN2;
                        const float xs[4] = {x.x, x.y, x.z, x.w};
                        float o[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool isp = q == tc;                         // st <= 0 makes tc negative
                            o[q] = focal_elem<BWD>(xs[q], isp, isp ? wp : wn);
                        }
                        if (BWD) dst4[v] = make_float4(o[0], o[1], o[2], o[3]);
                        else sums[0] += (o[0] + o[1]) + (o[2] + o[3]);
                    }
                }
            } else {
                for (int e = lane; e < nA * C; e += 64) {                      // any C: no prefetch, one value per lane
                    const int al = e / C;
                    const int c = e - al * C;
                    const int st = s_wstate[wv][al];
                    const bool isp = st > 0 && c == st - 1;
                    const float wn = st < 0 ? 0.f : (BWD ? -0.75f * cls_scale : 0.75f * RN_LN2);
                    const float wp = BWD ? 0.25f * cls_scale : 0.25f * RN_LN2;
                    const float o = focal_elem<BWD>(src[e], isp, isp ? wp : wn);
                    if (BWD) dst[e] = o;
                    else sums[0] += o;
                }
            }
        }

        // backward: zero the unit's dense dreg rows now; the rows of its positives are written when the queue is
        // drained, by the same wave (same-wave stores to one address stay in order)
        if (BWD) {
            float4 *z = reinterpret_cast<float4 *>(dreg + ((int64_t)j * A + a0) * NREG);
            const int nz = nA * (NREG / 4);
            for (int v = lane; v < nz; v += 64) z[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();                                       // the state row is reused by the next unit
    };

    // ---- (3) positives: regression / VP terms (forward) or their gradient rows (backward), one queued anchor per
    // lane.  Runs when no prefetch is in flight (these terms are register-hungry), on the positives of several units
    // at once.
    auto drain = [&]() {
        for (int q = lane; q < qn; q += 64) {
            const int64_t ai = s_q[0][wv][q];
            const float *gl = s_raw + s_q[1][wv][q] * COLS;
            const float *r = reg + ((int64_t)j * A + ai) * NREG;
            float *dr = BWD ? dreg + ((int64_t)j * A + ai) * NREG : nullptr;
            sums[3] += 1.f;
            if (DIR) positive_dir<BWD>(anchors[ai], gl, r, sums[1], sums[2], reg_scale, vp_scale, dr);
            else positive_2d<BWD>(anchors[ai], gl, r, sums[1], reg_scale, dr);
        }
        __builtin_amdgcn_wave_barrier();
        qn = 0;
    };

    bool primed = true;
    while (u < units) {                                                        // wave-uniform; normally a single pass
        if (!primed) issue(u, avA, xvA);
        primed = false;
        while (true) {                                                         // invariant: qn <= QCAP - UNIT
            int un = u + stride;
            issue(un < units ? un : u, avB, xvB);                              // past the end: re-request the current unit (L2-hot)
            process(u, avA, xvA);
            u = un;
            if (u >= units || qn > QCAP - UNIT) break;                         // queue nearly full: drop the prefetch, drain, resume
            un = u + stride;
            issue(un < units ? un : u, avA, xvA);
            process(u, avB, xvB);
            u = un;
            if (u >= units || qn > QCAP - UNIT) break;
        }
        drain();
    }
    if (!BWD) {
        sums[0] += (-0.75f * RN_LN2) * (neg2.x + neg2.y);
        block_sum4<NWAVES>(sums, s_red);
        if (threadIdx.x == 0) {
            // Publish the partial with agent-scope atomic exchanges: they are performed at the memory-side coherence
            // point, visible to every XCD, without an L2 writeback (a __threadfence() per workgroup is a full L2 flush
            // on this multi-XCD part and costs more than the whole kernel).  The count goes out once they are done.
            unsigned long long *p64 = reinterpret_cast<unsigned long long *>(partials) + 2 * ((int64_t)j * G + g);
            const unsigned long long lo = ((unsigned long long)__float_as_uint(sums[1]) << 32) | __float_as_uint(sums[0]);
            const unsigned long long hi = ((unsigned long long)__float_as_uint(sums[3]) << 32) | __float_as_uint(sums[2]);
            (void)__hip_atomic_exchange(p64, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_exchange(p64 + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_last = atomicAdd(counter + RN_WS_HEAD / 4 + j, 1u) == (unsigned)G - 1;   // this image's counter: G arrivals
        }
        __syncthreads();
        if (s_last) {                                                          // every partial of image j is in
            double *s_red = reinterpret_cast<double *>(&s_q[0][0][0]);
            const int B = (int)(gridDim.x / G);
            focal_finalize_image<DIR>(reinterpret_cast<const float4 *>(partials), G, j, ann, N, terms, s_red);
            if (threadIdx.x == 0) {
                __hip_atomic_store(counter + RN_WS_HEAD / 4 + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning
                s_last = atomicAdd(counter, 1u) == (unsigned)B - 1;            // batch counter: B arrivals
            }
            __syncthreads();
            if (s_last) {                                                      // every image's terms are in
                focal_finalize_batch<DIR>(B, terms, stats, losses, s_red, s_red + NWAVES * 4);
                if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

static int check_args(int B, int64_t A, int C, int N) {
    if (B <= 0 || A <= 0 || C <= 0 || N < 0) return RN_EINVAL;
    if (N > RN_MAX_GT) return RN_ETOOMANY;
    if (B > 1024) return RN_EINVAL;
    if (A > 0x7fffffffLL) return RN_EINVAL;                                      // anchor / unit ids are ints; grid = B*G <= 1024*160
    return RN_OK;
}

struct FocalWs {
    unsigned *counter;
    ImageStats *stats;
    double *terms;
    float *partials;
};
static inline FocalWs focal_ws(void *workspace, int B) {
    char *w = reinterpret_cast<char *>(workspace);
    FocalWs r;
    r.counter = reinterpret_cast<unsigned *>(w);
    const size_t head = RN_WS_HEAD + (size_t)focal_counter_bytes(B);
    r.stats = reinterpret_cast<ImageStats *>(w + head);
    r.terms = reinterpret_cast<double *>(w + head + (size_t)B * sizeof(ImageStats));
    r.partials = reinterpret_cast<float *>(w + head + (size_t)B * sizeof(ImageStats) + (size_t)B * RN_TERMS * sizeof(double));
    return r;
}

extern "C" int rn_focal_loss_fwd(const float *cls, const float *reg, const float *anchors, const float *ann, int B,
                                 int64_t A, int C, int N, int directional, void *workspace, float *losses,
                                 void *stream) {
    const int rc = check_args(B, A, C, N);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int G = focal_groups(B, A);
    const int units = (int)((A + UNIT - 1) / UNIT);
    const FocalWs ws = focal_ws(workspace, B);
    const dim3 grid((unsigned)(B * G)), block(NTHR);
    const float4 *anc = reinterpret_cast<const float4 *>(anchors);
    const int cq = C == 4 ? 1 : (C == 8 ? 2 : 0);
    const size_t raw_bytes = (size_t)(N > 0 ? N : 1) * (directional ? 27 : 5) * sizeof(float);   // label rows in LDS
#define RN_FOCAL_FWD(DIR_, CQ_)                                                                                        \
    hipLaunchKernelGGL((focal_kernel<DIR_, false, CQ_>), grid, block, raw_bytes, s, cls, reg, anc, ann, A, C, N,        \
                       ws.partials, ws.stats, (float *)nullptr, (float *)nullptr, G, units, ws.counter, ws.terms, losses, \
                       (const float *)nullptr)
    if (directional) {
        if (cq == 2) RN_FOCAL_FWD(true, 2); else if (cq == 1) RN_FOCAL_FWD(true, 1); else RN_FOCAL_FWD(true, 0);
    } else {
        if (cq == 2) RN_FOCAL_FWD(false, 2); else if (cq == 1) RN_FOCAL_FWD(false, 1); else RN_FOCAL_FWD(false, 0);
    }
#undef RN_FOCAL_FWD
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_focal_loss_bwd(const float *cls, const float *reg, const float *anchors, const float *ann, int B,
                                 int64_t A, int C, int N, int directional, const void *workspace,
                                 const float *grad_losses, float *dcls, float *dreg, void *stream) {
    const int rc = check_args(B, A, C, N);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int G = focal_groups(B, A);
    const int units = (int)((A + UNIT - 1) / UNIT);
    const FocalWs ws = focal_ws(const_cast<void *>(workspace), B);
    const dim3 grid((unsigned)(B * G)), block(NTHR);
    const float4 *anc = reinterpret_cast<const float4 *>(anchors);
    const int cq = C == 4 ? 1 : (C == 8 ? 2 : 0);
    const size_t raw_bytes = (size_t)(N > 0 ? N : 1) * (directional ? 27 : 5) * sizeof(float);   // label rows in LDS
#define RN_FOCAL_BWD(DIR_, CQ_)                                                                                      \
    hipLaunchKernelGGL((focal_kernel<DIR_, true, CQ_>), grid, block, raw_bytes, s, cls, reg, anc, ann, A, C, N,       \
                       (float *)nullptr, ws.stats, dcls, dreg, G, units, (unsigned *)nullptr, (double *)nullptr,      \
                       (float *)nullptr, grad_losses)
    if (directional) {
        if (cq == 2) RN_FOCAL_BWD(true, 2); else if (cq == 1) RN_FOCAL_BWD(true, 1); else RN_FOCAL_BWD(true, 0);
    } else {
        if (cq == 2) RN_FOCAL_BWD(false, 2); else if (cq == 1) RN_FOCAL_BWD(false, 1); else RN_FOCAL_BWD(false, 0);
    }
#undef RN_FOCAL_BWD
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
