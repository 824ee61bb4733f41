// Box decode, clipping, score filtering and NMS -- the eval branch of the detector.
//
// Replaces BBoxTransform.forward (D/utils.py:102-149 directional, R/utils.py:102-126 2D), ClipBoxes.forward
// (R/utils.py:134-144), the adaptive-threshold while-loops and per-class / batched NMS of ResNet.forward
// (D/model.py:311-397, R/model.py:283-311) and torchvision.ops.nms (third party; contract restated in
// oracle/boxes.py, parity unpinned).
//
// Compiled with -ffp-contract=off: decode and IoU are evaluated with one rounding per operation in the
// reference's order, so survivors and kept indices are bit-identical to the CPU path.
//
// Roofline: HBM for decode (read B*A*12*4 + A*16, write B*A*20*4); the filter is one read of the scores; NMS
// works on <= 10 000 candidates (D/model.py:368) and is latency-bound.
#include <math.h>
#include <cmath>

#include "common.h"

// ------------------------------------------------------------------------------------------------ decode
// One anchor's decode: 12 regression values -> 16 corner coordinates + 2D box (D/utils.py:104-135), written as 5 float4.
__device__ __forceinline__ void decode_dir_one(const float4 a, const float *__restrict__ reg12, float *__restrict__ out20) {
    const float w = a.z - a.x, h = a.w - a.y;                                  // D/utils.py:104-107
    const float cx = a.x + 0.5f * w, cy = a.y + 0.5f * h;
    const float4 *r4 = reinterpret_cast<const float4 *>(reg12);
    const float4 r0 = r4[0], r1 = r4[1], r2 = r4[2];
    const float r[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
    float o[20];
    const float SL[8] = {-1, -1, +1, +1, -1, -1, +1, +1};                      // D/utils.py:113-130
    const float SW[8] = {-1, +1, -1, +1, -1, +1, -1, +1};
    const float SH[8] = {+1, +1, +1, +1, -1, -1, -1, -1};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float px = ((r[0] + SL[j] * r[2]) + SW[j] * r[4]) + SH[j] * r[6];
        const float py = ((r[1] + SL[j] * r[3]) + SW[j] * r[5]) + SH[j] * r[7];
        o[2 * j] = px * w + cx;                                                // D/utils.py:134
        o[2 * j + 1] = py * h + cy;                                            // D/utils.py:135
    }
    o[16] = r[8] * w + cx;
    o[17] = r[9] * h + cy;
    o[18] = r[10] * w + cx;
    o[19] = r[11] * h + cy;
    float4 *o4 = reinterpret_cast<float4 *>(out20);
#pragma unroll
    for (int k = 0; k < 5; ++k) o4[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

__global__ __launch_bounds__(256) void decode_dir_kernel(const float4 *__restrict__ anchors, const float *__restrict__ reg,
                                                         float *__restrict__ out, int64_t A, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    decode_dir_one(anchors[i % A], reg + i * 12, out + i * 20);
}

// Decode of the score filter's survivors only: candidate k = flat anchor index sel[k] of [B*A].  The eval
// branches decode all B*A anchors first (D/model.py:347: 249 MB written at batch 8) and then keep <= 10 000 of them; the
// arithmetic per anchor is the same function, so the kept boxes are bit-identical.  Also emits what the NMS and the
// output gather need in candidate order: the candidate's score and its image index (D/model.py:314-316).
__global__ __launch_bounds__(256) void decode_dir_select_kernel(const float4 *__restrict__ anchors, const float *__restrict__ reg,
                                                                int64_t A, const float *__restrict__ scores, int64_t score_stride,
                                                                const int32_t *__restrict__ sel, const int32_t *__restrict__ count,
                                                                int max_cand, float *__restrict__ boxes, float *__restrict__ cscore,
                                                                int32_t *__restrict__ image) {
    int n = count[0];
    if (n > max_cand) n = max_cand;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int64_t i = sel[k];
    decode_dir_one(anchors[i % A], reg + i * 12, boxes + (int64_t)k * 20);
    cscore[k] = scores[i * score_stride];
    if (image) image[k] = (int32_t)(i / A);
}

extern "C" int rn_decode_dir(const float *anchors, const float *reg, float *boxes, int B, int64_t A, void *stream) {
    if (B <= 0 || A <= 0) return RN_EINVAL;
    const int64_t total = (int64_t)B * A;
    hipLaunchKernelGGL(decode_dir_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(anchors), reg, boxes, A, total);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ __launch_bounds__(256) void decode_2d_kernel(const float4 *__restrict__ anchors, const float4 *__restrict__ deltas,
                                                        float4 *__restrict__ out, int64_t A, int64_t total, int clip,
                                                        float width, float height) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float4 a = anchors[i % A];
    const float4 d = deltas[i];
    const float w = a.z - a.x, h = a.w - a.y;
    const float cx = a.x + 0.5f * w, cy = a.y + 0.5f * h;
    const float dx = d.x * 0.1f + 0.f, dy = d.y * 0.1f + 0.f;                  // R/utils.py:109-112 (std, mean 0)
    const float dw = d.z * 0.2f + 0.f, dh = d.w * 0.2f + 0.f;
    const float pcx = cx + dx * w, pcy = cy + dy * h;                          // R/utils.py:114-117
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
    if (clip) {                                                                // R/utils.py:138-142
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f);
        o.z = fminf(o.z, width); o.w = fminf(o.w, height);
    }
    out[i] = o;
}

extern "C" int rn_decode_dir_select(const float *anchors, const float *reg, int64_t A, const float *scores, int64_t score_stride,
                                    const int32_t *sel_idx, const int32_t *count, int max_candidates, float *boxes,
                                    float *cand_scores, int32_t *cand_image, void *stream) {
    if (A <= 0 || max_candidates <= 0 || score_stride <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(decode_dir_select_kernel, dim3(rn_blocks(max_candidates, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(anchors), reg, A, scores, score_stride, sel_idx, count, max_candidates,
                       boxes, cand_scores, cand_image);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_decode_2d(const float *anchors, const float *deltas, float *boxes, int B, int64_t A, int clip,
                            float width, float height, void *stream) {
    if (B <= 0 || A <= 0) return RN_EINVAL;
    const int64_t total = (int64_t)B * A;
    hipLaunchKernelGGL(decode_2d_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(anchors), reinterpret_cast<const float4 *>(deltas),
                       reinterpret_cast<float4 *>(boxes), A, total, clip, width, height);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

__global__ void clip_kernel(float4 *__restrict__ b, int64_t n, float width, float height) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 o = b[i];
    o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f);
    o.z = fminf(o.z, width); o.w = fminf(o.w, height);
    b[i] = o;
}

extern "C" int rn_clip_boxes(float *boxes, int64_t n, float width, float height, void *stream) {
    if (n <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(clip_kernel, dim3(rn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4 *>(boxes), n, width, height);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ row max
__global__ void rowmax_kernel(const float *__restrict__ cls, int64_t n, int C, float *__restrict__ scores,
                              int64_t *__restrict__ classes) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = cls + i * C;
    float best = p[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = p[c];
        if (v > best) { best = v; arg = c; }                                   // first maximum, D/model.py:320
    }
    scores[i] = best;
    classes[i] = arg;
}

extern "C" int rn_rowmax(const float *cls, int64_t n, int C, float *scores, int64_t *classes, void *stream) {
    if (n <= 0 || C <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(rowmax_kernel, dim3(rn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, cls, n, C, scores, classes);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ adaptive threshold
#define MAX_THR 352           // 1e-25 * 10^(0.2 k) reaches fp32 +inf at k = 318
#define SEL_BLOCK 1024        // scores per compaction block

struct ThrTable {
    float t[MAX_THR];
    int count;
};

struct SelectWs {             // layout of the select workspace
    int hist[MAX_THR + 1];
    int chosen;               // index into ThrTable
    int total;                // number selected
    int pad[2];
    // followed by int block_count[nblocks], int block_offset[nblocks]
};

__device__ __forceinline__ int thr_bin(float s, const ThrTable &T) {           // #{k : s > t_k}, t ascending
    int lo = 0, hi = T.count;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s > T.t[mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void thr_hist_kernel(const float *__restrict__ scores, int64_t n, int64_t stride,
                                                       ThrTable T, SelectWs *__restrict__ ws) {
    __shared__ int h[MAX_THR + 1];
    for (int k = threadIdx.x; k <= T.count; k += 256) h[k] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        atomicAdd(&h[thr_bin(scores[i * stride], T)], 1);
    __syncthreads();
    for (int k = threadIdx.x; k <= T.count; k += 256)
        if (h[k]) atomicAdd(&ws->hist[k], h[k]);
}

// The reference's loop: smallest k with #{s > t_k} <= keep.
__device__ __forceinline__ int pick_threshold(const SelectWs *ws, int K, int keep, int *total) {
    int above = 0;
    for (int b = 1; b <= K; ++b) above += ws->hist[b];                         // #{s > t_0} = bins 1..K
    int k = 0;
    while (k < K - 1 && above > keep) {                                        // move to t_{k+1}: drop bin k+1
        above -= ws->hist[k + 1];
        ++k;
    }
    *total = above;
    return k;
}

__global__ __launch_bounds__(256) void thr_count_kernel(const float *__restrict__ scores, int64_t n, int64_t stride,
                                                        ThrTable T, int keep, SelectWs *__restrict__ ws,
                                                        int *__restrict__ block_count) {
    __shared__ int s_k, s_cnt[4];
    if (threadIdx.x == 0) {
        int total;
        s_k = pick_threshold(ws, T.count, keep, &total);
        if (blockIdx.x == 0) { ws->chosen = s_k; ws->total = total; }
    }
    __syncthreads();
    const float thr = T.t[s_k];
    const int64_t base = (int64_t)blockIdx.x * SEL_BLOCK;
    int c = 0;
#pragma unroll
    for (int q = 0; q < SEL_BLOCK / 256; ++q) {
        const int64_t i = base + q * 256 + threadIdx.x;
        c += (i < n && scores[i * stride] > thr) ? 1 : 0;
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
}

__global__ __launch_bounds__(1024) void thr_scan_kernel(const int *__restrict__ block_count, int nblocks,
                                                        int *__restrict__ block_offset, const SelectWs *__restrict__ ws,
                                                        int *__restrict__ out_count) {
    __shared__ int s[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const int v = b < nblocks ? block_count[b] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                             // Hillis-Steele inclusive scan
            const int add = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (b < nblocks) block_offset[b] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_count[0] = ws->total;
}

__global__ __launch_bounds__(256) void thr_scatter_kernel(const float *__restrict__ scores, int64_t n, int64_t stride,
                                                          ThrTable T, const SelectWs *__restrict__ ws,
                                                          const int *__restrict__ block_offset, int32_t *__restrict__ sel_idx) {
    __shared__ int s_wave[4];
    const float thr = T.t[ws->chosen];
    const int64_t base = (int64_t)blockIdx.x * SEL_BLOCK;
    int running = block_offset[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < SEL_BLOCK / 256; ++q) {
        const int64_t i = base + q * 256 + threadIdx.x;
        const bool sel = i < n && scores[i * stride] > thr;
        const unsigned long long m = __ballot(sel);
        if (lane == 0) s_wave[w] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int k = 0; k < w; ++k) before += s_wave[k];
        const int all = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
        if (sel) sel_idx[running + before + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
        running += all;
        __syncthreads();
    }
}

static int64_t select_ws_bytes(int64_t n) {
    const int64_t nblocks = (n + SEL_BLOCK - 1) / SEL_BLOCK;
    return (int64_t)sizeof(SelectWs) + 2 * nblocks * (int64_t)sizeof(int);
}

extern "C" int rn_threshold_select(const float *scores, int64_t n, int64_t stride, double start, int keep,
                                   double fixed_threshold, void *workspace, int32_t *count, int32_t *sel_idx,
                                   void *stream) {
    if (n <= 0 || stride <= 0 || n > 0x7fffffffLL) return RN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ThrTable T;
    if (fixed_threshold >= 0.0) {                                              // R/model.py:289: scores > 0.05
        T.t[0] = (float)fixed_threshold;
        T.count = 1;
        keep = 0x7fffffff;
    } else {
        double t = start;                                                      // D/model.py:324-328, 370-374
        int k = 0;
        for (; k < MAX_THR; ++k) {
            T.t[k] = (float)t;                                                 // python scalar compared in fp32
            if (std::isinf(T.t[k])) { ++k; break; }
            t *= pow(10.0, 0.2);                                               // threshold *= (10**.2)
        }
        T.count = k;
    }
    SelectWs *ws = reinterpret_cast<SelectWs *>(workspace);
    const int nblocks = (int)((n + SEL_BLOCK - 1) / SEL_BLOCK);
    int *block_count = reinterpret_cast<int *>(ws + 1);
    int *block_offset = block_count + nblocks;
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(SelectWs), s);
    if (e != hipSuccess) return (int)e;
    const int hist_blocks = nblocks < 2048 ? (nblocks * 4 < 1 ? 1 : (nblocks * 4 > 2048 ? 2048 : nblocks * 4)) : 2048;
    hipLaunchKernelGGL(thr_hist_kernel, dim3(hist_blocks), dim3(256), 0, s, scores, n, stride, T, ws);
    hipLaunchKernelGGL(thr_count_kernel, dim3(nblocks), dim3(256), 0, s, scores, n, stride, T, keep, ws, block_count);
    hipLaunchKernelGGL(thr_scan_kernel, dim3(1), dim3(1024), 0, s, (const int *)block_count, nblocks, block_offset,
                       (const SelectWs *)ws, count);
    hipLaunchKernelGGL(thr_scatter_kernel, dim3(nblocks), dim3(256), 0, s, scores, n, stride, T, (const SelectWs *)ws,
                       (const int *)block_offset, sel_idx);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ------------------------------------------------------------------------------------------------ NMS
#define NMS_MAX 16384         // candidates one sort workgroup can order in LDS (128 KiB of 8-byte keys)

struct NmsWs {
    float max_coord;
    int pad[3];
    // int32 order[max_cand] | float4 sorted_box[max_cand] | uint64 mask[max_cand][words]
};

static int64_t nms_ws_bytes(int64_t max_cand) {
    const int64_t words = (max_cand + 63) / 64;
    return 16 + max_cand * 4 + 16 /*align*/ + max_cand * 16 + max_cand * words * 8;
}

extern "C" int64_t rn_post_workspace_bytes(int64_t n_scores, int64_t max_candidates) {
    const int64_t a = select_ws_bytes(n_scores), b = nms_ws_bytes(max_candidates);
    return (a > b ? a : b) + 64;
}

__device__ __forceinline__ unsigned int f32_desc_key(float f) {                // larger float -> smaller key
    unsigned int u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                            // ascending in f
    return ~u;
}

// One workgroup: keys (score descending, candidate position ascending) sorted by bitonic network in LDS;
// also reduces the maximum coordinate of the candidate boxes (batched_nms, D/model.py:52).
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float *__restrict__ boxes, int64_t box_stride, int box_col,
                                                        const float *__restrict__ scores, int64_t score_stride,
                                                        const int32_t *__restrict__ cand_idx, const int32_t *__restrict__ count,
                                                        int max_cand, NmsWs *__restrict__ ws, int32_t *__restrict__ order) {
    __shared__ unsigned long long keys[NMS_MAX];
    __shared__ float s_max[16];
    int n = count[0];
    if (n > max_cand) n = max_cand;
    int npad = 64;
    while (npad < n) npad <<= 1;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < npad; i += 1024) {
        unsigned long long k = ~0ull;
        if (i < n) {
            const int64_t src = cand_idx[i];
            k = ((unsigned long long)f32_desc_key(scores[src * score_stride]) << 32) | (unsigned int)i;
            const float *b = boxes + src * box_stride + box_col;
            mx = fmaxf(mx, fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
        }
        keys[i] = k;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_max[0];
        for (int k = 1; k < 16; ++k) m = fmaxf(m, s_max[k]);
        ws->max_coord = m;
    }
    for (int size = 2; size <= npad; size <<= 1) {
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (npad >> 1); t += 1024) {
                const int lo = 2 * t - (t & (strd - 1));                       // index with bit `strd` clear
                const int hi = lo + strd;
                const bool up = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) order[i] = (int32_t)(keys[i] & 0xffffffffu);
}

__global__ __launch_bounds__(256) void nms_gather_kernel(const float *__restrict__ boxes, int64_t box_stride, int box_col,
                                                         const int32_t *__restrict__ cand_idx, const int32_t *__restrict__ category,
                                                         const int32_t *__restrict__ count, int max_cand,
                                                         const NmsWs *__restrict__ ws, const int32_t *__restrict__ order,
                                                         float4 *__restrict__ sorted_box) {
    int n = count[0];
    if (n > max_cand) n = max_cand;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int ci = order[p];
    const float *b = boxes + (int64_t)cand_idx[ci] * box_stride + box_col;
    float4 o = make_float4(b[0], b[1], b[2], b[3]);
    if (category) {                                                            // D/model.py:52-55
        const float off = (float)category[ci] * (ws->max_coord + 1.0f);
        o.x += off; o.y += off; o.z += off; o.w += off;
    }
    sorted_box[p] = o;
}

// mask[i][w] bit b = 1 iff IoU(box i, box 64w+b) > thr and 64w+b > i.  One wave per (row block, column block).
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4 *__restrict__ sorted_box, const int32_t *__restrict__ count,
                                                      int max_cand, int words, float thr, unsigned long long *__restrict__ mask) {
    int n = count[0];
    if (n > max_cand) n = max_cand;
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (rb * 64 >= n || cb * 64 >= n || cb < rb) return;
    __shared__ float4 cbox[64];
    const int cj = cb * 64 + threadIdx.x;
    cbox[threadIdx.x] = cj < n ? sorted_box[cj] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= n) return;
    const float4 a = sorted_box[i];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    unsigned long long bits = 0ull;
    const int lim = (n - cb * 64) < 64 ? (n - cb * 64) : 64;
    for (int k = 0; k < lim; ++k) {
        const int jdx = cb * 64 + k;
        if (jdx <= i) continue;
        const float4 b = cbox[k];
        const float iw = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
        const float ih = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
        const float inter = iw * ih;
        const float area_b = (b.z - b.x) * (b.w - b.y);
        const float iou = inter / ((area_a + area_b) - inter);
        if (iou > thr) bits |= 1ull << k;
    }
    mask[(int64_t)i * words + cb] = bits;
}

// Sequential part: chunks of 64 sorted candidates.  Thread t owns word t of the "removed" bitset
// (NMS_MAX / 64 = 256 words).  Wave 0 resolves a chunk against its diagonal mask words, then every thread ORs
// in the rows the chunk kept.
__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long *__restrict__ mask, int words,
                                                       const int32_t *__restrict__ count, int max_cand,
                                                       const int32_t *__restrict__ order, int32_t *__restrict__ keep,
                                                       int32_t *__restrict__ keep_count) {
    __shared__ unsigned long long s_removed[NMS_MAX / 64];
    __shared__ unsigned long long s_keepbits;
    int n = count[0];
    if (n > max_cand) n = max_cand;
    const int t = threadIdx.x;
    unsigned long long removed = 0ull;
    int nkeep = 0;
    const int chunks = (n + 63) / 64;
    // The diagonal word of a chunk's rows depends on nothing the scan computes: it is fetched one chunk ahead, so that of the
    // two global-memory round trips a chunk used to pay in sequence (diagonal, then the kept rows) only the second remains.
    auto diag_of = [&](int c) -> unsigned long long {
        const int row = c * 64 + t;
        return (t < 64 && c < chunks && row < n) ? mask[(int64_t)row * words + c] : 0ull;
    };
    unsigned long long diag_next = diag_of(0);
    for (int c = 0; c < chunks; ++c) {
        const unsigned long long diag = diag_next;
        diag_next = diag_of(c + 1);
        s_removed[t] = removed;
        __syncthreads();
        if (t < 64) {
            unsigned long long cur = s_removed[c];
            unsigned long long kb = 0ull;
            const int lim = (n - c * 64) < 64 ? (n - c * 64) : 64;
            for (int r = 0; r < lim; ++r) {
                const unsigned long long d = __shfl(diag, r, 64);
                if (!((cur >> r) & 1ull)) { kb |= 1ull << r; cur |= d; }
            }
            if (t == 0) s_keepbits = kb;
            if (((kb >> t) & 1ull)) {
                const int pos = nkeep + __popcll(kb & ((1ull << t) - 1ull));
                keep[pos] = order[c * 64 + t];
            }
        }
        __syncthreads();
        const unsigned long long kb = s_keepbits;
        nkeep += __popcll(kb);
        if (t > c && t < words) {
            // the kept rows' words, four loads in flight at a time
            unsigned long long a0 = 0ull, a1 = 0ull, a2 = 0ull, a3 = 0ull;
            unsigned long long bits = kb;
            while (bits) {
                int r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    r[k] = bits ? __ffsll((long long)bits) - 1 : -1;
                    bits &= bits - 1;                         // (0 stays 0)
                }
                const unsigned long long v0 = mask[(int64_t)(c * 64 + r[0]) * words + t];
                const unsigned long long v1 = r[1] >= 0 ? mask[(int64_t)(c * 64 + r[1]) * words + t] : 0ull;
                const unsigned long long v2 = r[2] >= 0 ? mask[(int64_t)(c * 64 + r[2]) * words + t] : 0ull;
                const unsigned long long v3 = r[3] >= 0 ? mask[(int64_t)(c * 64 + r[3]) * words + t] : 0ull;
                a0 |= v0; a1 |= v1; a2 |= v2; a3 |= v3;
            }
            removed |= (a0 | a1) | (a2 | a3);
        }
        __syncthreads();
    }
    if (t == 0) keep_count[0] = nkeep;
}

extern "C" int rn_nms(const float *boxes, int64_t box_stride, int box_col, const float *scores, int64_t score_stride,
                      const int32_t *cand_idx, const int32_t *category, const int32_t *count, int max_candidates,
                      float iou_thr, void *workspace, int32_t *keep, int32_t *keep_count, void *stream) {
    if (max_candidates <= 0 || max_candidates > NMS_MAX) return RN_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int words = (max_candidates + 63) / 64;
    NmsWs *ws = reinterpret_cast<NmsWs *>(workspace);
    char *p = reinterpret_cast<char *>(workspace) + 16;
    int32_t *order = reinterpret_cast<int32_t *>(p);
    p += (int64_t)max_candidates * 4;
    p = reinterpret_cast<char *>(((uintptr_t)p + 15) & ~(uintptr_t)15);
    float4 *sorted_box = reinterpret_cast<float4 *>(p);
    p += (int64_t)max_candidates * 16;
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(p);
    hipLaunchKernelGGL(nms_sort_kernel, dim3(1), dim3(1024), 0, s, boxes, box_stride, box_col, scores, score_stride,
                       cand_idx, count, max_candidates, ws, order);
    hipLaunchKernelGGL(nms_gather_kernel, dim3(rn_blocks(max_candidates, 256)), dim3(256), 0, s, boxes, box_stride, box_col,
                       cand_idx, category, count, max_candidates, (const NmsWs *)ws, (const int32_t *)order, sorted_box);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, s, (const float4 *)sorted_box, count,
                       max_candidates, words, iou_thr, mask);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), 0, s, (const unsigned long long *)mask, words, count,
                       max_candidates, (const int32_t *)order, keep, keep_count);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
