// Detection parsing of the multi-camera tracker, device-resident.
//
// Replaces MC_Crop_Tracker.parse_detections with im_nms and space_nms (MC3D_crop_tracker.py:319-383, 592-636): the
// step right after the MULTI_FRAME detector.  The reference copies the detector's four outputs to the host
// (MC3D_crop_tracker.py:1078-1083) and runs, per frame set, a boolean-mask filter, an NMS on image envelopes, a
// Python list comprehension of camera names feeding per-object float64 bmm homographies (twice more when heights
// are refined), a second NMS on road-plane footprints and four fancy-index gathers -- all on the CPU.
//
// Here the chain never leaves the device and never synchronises: every stage reads its element count from device
// memory and the last stage leaves (state, labels, scores, cameras, count) compacted in NMS order.
//   tp_filter_kernel   scores > sigma_d, ordered compaction (a boolean mask keeps input order), image envelope of the 8
//                      corners shifted by the constant 10 000 -- the reference computes a per-camera offset and then adds
//                      the same constant to every box (MC3D_crop_tracker.py:610-613), so cameras do NOT separate
//                      detections; the IoU is evaluated on the shifted fp32 coordinates, as there
//   rn_nms             (boxes.hip) greedy NMS, IoU > phi_nms_im
//   tp_state_kernel    one lane per survivor: image -> state through the camera's H (both homographies of the wrapper,
//                      switch at y > 60), optional height refinement (state -> image through P, height_from_template,
//                      image -> state again), road-plane footprint of the result; fp64 projection arithmetic as in
//                      homography.py, operation order kept
//   rn_nms             IoU > phi_nms_space on the footprints
//   tp_gather_kernel   survivors in NMS order
// Compiled with -ffp-contract=off: NMS decisions must match the CPU path bit for bit.
// Latency-bound by construction (hundreds to a few thousand detections): report microseconds per call.
#include <math.h>

#include "common.h"
#include "homography_dev.h"

struct ParseCounts {
    int32_t n1;   // above sigma_d
    int32_t n2;   // after the image NMS (= n1 when NMS is off)
    int32_t n3;   // after the space NMS (= n2 when NMS is off)
    int32_t pad;
};

struct ParseWs {
    ParseCounts *cnt;
    int32_t *sel, *iota, *keep1, *src2, *keep2;
    float *sc1, *sc2, *st2;
    float4 *env1, *foot2;
    void *nms;
};

static inline int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }

static int64_t parse_ws_layout(void *base, int64_t d, int maxc, ParseWs *w) {
    char *p = reinterpret_cast<char *>(base);
    int64_t o = 0;
    auto take = [&](int64_t bytes) { char *r = p ? p + o : nullptr; o += align16(bytes); return r; };
    ParseWs t;
    t.cnt = reinterpret_cast<ParseCounts *>(take(sizeof(ParseCounts)));
    t.sel = reinterpret_cast<int32_t *>(take(d * 4));
    t.iota = reinterpret_cast<int32_t *>(take(d * 4));
    t.keep1 = reinterpret_cast<int32_t *>(take(d * 4));
    t.src2 = reinterpret_cast<int32_t *>(take(d * 4));
    t.keep2 = reinterpret_cast<int32_t *>(take(d * 4));
    t.sc1 = reinterpret_cast<float *>(take(d * 4));
    t.sc2 = reinterpret_cast<float *>(take(d * 4));
    t.st2 = reinterpret_cast<float *>(take(d * 24));
    t.env1 = reinterpret_cast<float4 *>(take(d * 16));
    t.foot2 = reinterpret_cast<float4 *>(take(d * 16));
    t.nms = take(rn_post_workspace_bytes(0, maxc));
    if (w) *w = t;
    return o;
}

extern "C" int64_t rn_parse_workspace_bytes(int64_t d) {
    if (d <= 0) return 0;
    const int maxc = (int)(d < RN_PARSE_MAX ? d : RN_PARSE_MAX);
    return parse_ws_layout(nullptr, d, maxc, nullptr);
}

// One workgroup walks the detections in order (a boolean mask keeps input order: MC3D_crop_tracker.py:339-344).
__global__ __launch_bounds__(1024) void tp_filter_kernel(const float *__restrict__ scores, const float *__restrict__ boxes20,
                                                         int64_t d, float sigma, float offset, ParseWs w) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < d; b0 += 1024) {
        const int64_t i = b0 + threadIdx.x;
        const float sc = i < d ? scores[i] : 0.f;
        const bool keep = i < d && sc > sigma;                                  // scores > ones * sigma_d
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = s_base;
        for (int k = 0; k < wave; ++k) before += s_wave[k];
        if (keep) {
            const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
            const float *b = boxes20 + i * 20;                                  // 8 corners (x,y) -- the 2D box is dropped
            float x1 = b[0], y1 = b[1], x2 = b[0], y2 = b[1];
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                x1 = fminf(x1, b[2 * k]); x2 = fmaxf(x2, b[2 * k]);
                y1 = fminf(y1, b[2 * k + 1]); y2 = fmaxf(y2, b[2 * k + 1]);
            }
            w.sel[pos] = (int32_t)i;
            w.iota[pos] = pos;
            w.sc1[pos] = sc;
            w.env1[pos] = make_float4(x1 + offset, y1 + offset, x2 + offset, y2 + offset);   // MC3D_crop_tracker.py:613
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = s_base;
            for (int k = 0; k < 16; ++k) t += s_wave[k];
            s_base = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        w.cnt->n1 = s_base;
        w.cnt->n2 = s_base;            // overwritten by the image NMS when it runs
        w.cnt->n3 = s_base;
    }
}

__global__ __launch_bounds__(256) void tp_state_kernel(const float *__restrict__ boxes20, const int64_t *__restrict__ camera_idxs,
                                                       const float *__restrict__ heights, const double *__restrict__ H1,
                                                       const double *__restrict__ H2, const double *__restrict__ P1,
                                                       const double *__restrict__ P2, int n_cam, int nms_im, int refine,
                                                       ParseWs w) {
    const int n = nms_im ? w.cnt->n2 : w.cnt->n1;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int j = nms_im ? w.keep1[k] : k;
    const int i = w.sel[j];
    int cam = (int)camera_idxs[i];
    cam = cam < 0 ? 0 : (cam >= n_cam ? n_cam - 1 : cam);                        // the reference would raise IndexError
    const float *b = boxes20 + (int64_t)i * 20;
    float px[8], py[8];
    double2 pt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { px[q] = b[2 * q]; py[q] = b[2 * q + 1]; pt[q] = make_double2((double)px[q], (double)py[q]); }
    const float h0 = heights ? heights[i] : 5.0f;                                // guess_heights: "other" (homography.py:514)
    float st[6];
    hg_im_to_state_refined<float>(pt, px, py, h0, refine != 0, H1, H2, P1, P2, cam, st);   // MC3D_crop_tracker.py:364-370
    float *o = w.st2 + (int64_t)k * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[q] = st[q];
    w.sc2[k] = w.sc1[j];
    w.src2[k] = i;
    w.iota[k] = k;                                                               // identity list for the second NMS
    w.foot2[k] = hg_footprint(st);                                                // for space_nms (MC3D_crop_tracker.py:626-633)
}

__global__ __launch_bounds__(256) void tp_gather_kernel(const int64_t *__restrict__ labels, const int64_t *__restrict__ camera_idxs,
                                                        int nms_im, int nms_space, ParseWs w, float *__restrict__ out_state,
                                                        int64_t *__restrict__ out_labels, float *__restrict__ out_scores,
                                                        int64_t *__restrict__ out_cams, int32_t *__restrict__ out_count) {
    const int n = nms_space ? w.cnt->n3 : (nms_im ? w.cnt->n2 : w.cnt->n1);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0) out_count[0] = n;
    if (k >= n) return;
    const int j = nms_space ? w.keep2[k] : k;
    const float *s = w.st2 + (int64_t)j * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) out_state[(int64_t)k * 6 + q] = s[q];
    out_scores[k] = w.sc2[j];
    const int i = w.src2[j];
    out_labels[k] = labels[i];
    out_cams[k] = camera_idxs[i];
}

extern "C" int rn_parse_detections(const float *scores, const int64_t *labels, const float *boxes20,
                                   const int64_t *camera_idxs, int64_t d, const double *H1, const double *H2,
                                   const double *P1, const double *P2, int n_cam, const float *heights, float sigma_d,
                                   float phi_nms_im, float phi_nms_space, int perform_nms, int refine_height,
                                   void *workspace, float *out_state, int64_t *out_labels, float *out_scores,
                                   int64_t *out_cams, int32_t *out_count, void *stream) {
    if (d <= 0 || n_cam <= 0 || !H1 || !workspace) return RN_EINVAL;
    if (refine_height && !P1) return RN_EINVAL;
    if (d > RN_PARSE_MAX) return RN_ETOOMANY;                       // the NMS orders its candidates in LDS
    hipStream_t s = (hipStream_t)stream;
    const int maxc = (int)d;
    ParseWs w;
    parse_ws_layout(workspace, d, maxc, &w);
    const int blocks = rn_blocks(d, 256);
    hipLaunchKernelGGL(tp_filter_kernel, dim3(1), dim3(1024), 0, s, scores, boxes20, d, sigma_d, 10000.0f, w);
    RN_LAUNCH_CHECK();
    const int nms_im = perform_nms & 1, nms_space = perform_nms & 2;
    if (nms_im) {
        const int rc = rn_nms(reinterpret_cast<const float *>(w.env1), 4, 0, w.sc1, 1, w.iota, nullptr, &w.cnt->n1, maxc,
                              phi_nms_im, w.nms, w.keep1, &w.cnt->n2, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(tp_state_kernel, dim3(blocks), dim3(256), 0, s, boxes20, camera_idxs, heights, H1, H2, P1, P2, n_cam,
                       nms_im, refine_height, w);
    RN_LAUNCH_CHECK();
    if (nms_space) {
        const int rc = rn_nms(reinterpret_cast<const float *>(w.foot2), 4, 0, w.sc2, 1, w.iota, nullptr,
                              nms_im ? &w.cnt->n2 : &w.cnt->n1, maxc,
                              phi_nms_space, w.nms, w.keep2, &w.cnt->n3, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(tp_gather_kernel, dim3(blocks), dim3(256), 0, s, labels, camera_idxs, nms_im, nms_space, w, out_state,
                       out_labels, out_scores, out_cams, out_count);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// md_iou (MC3D_crop_tracker.py:1030-1049): element-wise IoU of two box arrays in fp64, no clamp on the union
// (0/0 -> NaN, as torch.div gives).
__global__ __launch_bounds__(256) void md_iou_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                     double *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *p = a + i * 4, *q = b + i * 4;
    const double area_a = (p[2] - p[0]) * (p[3] - p[1]);
    const double area_b = (q[2] - q[0]) * (q[3] - q[1]);
    const double minx = fmax(p[0], q[0]), maxx = fmin(p[2], q[2]);
    const double miny = fmax(p[1], q[1]), maxy = fmin(p[3], q[3]);
    const double inter = fmax(0.0, maxx - minx) * fmax(0.0, maxy - miny);
    out[i] = inter / ((area_a + area_b) - inter);
}

extern "C" int rn_md_iou(const double *a, const double *b, double *out, int64_t n, void *stream) {
    if (n <= 0) return RN_EINVAL;
    hipLaunchKernelGGL(md_iou_kernel, dim3(rn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
