// Crop-refinement path of the multi-camera tracker, device-resident.
//
// Replaces, around the LOCALIZE detector call of MC_Crop_Tracker.track (MC3D_crop_tracker.py:1172-1226):
//   rn_crop_boxes    get_crop_boxes (:920-944) + the RoI rows of :1183-1184: square crops of side max(w,h)*b around the
//                    image envelope of each predicted object (float64, as the float64 state_to_im output makes them)
//   rn_roi_align     torchvision.ops.roi_align(frames, rois, (cs,cs)) of :1185 -- spatial_scale 1, sampling_ratio -1
//                    (adaptive grid), aligned=False, fp32, torchvision's operation order; output NCHW like torchvision or
//                    NHWC4 for the stem convolution directly
//   rn_crop_select   everything after the detector (:1192-1226): max over classes, local_to_global (float32 detection x
//                    float64 crop scale -> float64 frame coordinates), top-cd_max by confidence, image -> state with the
//                    height refinement through the object's camera, road-plane footprint IoU (float64) against the
//                    prior, score (1-W)*IoU + W*conf, first maximum -- one workgroup per object, nothing leaves the
//                    device; the reference copies four tensors to the host and runs this in torch CPU ops.
// Compiled with -ffp-contract=off (exact-arithmetic list).
#include <math.h>

#include "common.h"
#include "homography_dev.h"

// ---------------------------------------------------------------------------------------------------- crop boxes
__global__ __launch_bounds__(256) void crop_boxes_kernel(const double *__restrict__ im, const int64_t *__restrict__ cam,
                                                         int n, double b, double *__restrict__ boxes,
                                                         float *__restrict__ rois) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *p = im + (int64_t)i * 16;
    double minx = p[0], maxx = p[0], miny = p[1], maxy = p[1];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        minx = fmin(minx, p[2 * k]); maxx = fmax(maxx, p[2 * k]);
        miny = fmin(miny, p[2 * k + 1]); maxy = fmax(maxy, p[2 * k + 1]);
    }
    const double w = maxx - minx, h = maxy - miny;
    const double scale = fmax(w, h) * b;                                       // :933
    const double cx = (minx + maxx) / 2.0, cy = (miny + maxy) / 2.0, hs = scale / 2.0;
    const double x1 = cx - hs, x2 = cx + hs, y1 = cy - hs, y2 = cy + hs;       // :936-939
    double *o = boxes + (int64_t)i * 4;
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
    if (rois) {                                                                // cat((cidx, crop_boxes)).float(), :1183-1185
        float *r = rois + (int64_t)i * 5;
        r[0] = (float)(double)cam[i]; r[1] = (float)x1; r[2] = (float)y1; r[3] = (float)x2; r[4] = (float)y2;
    }
}

extern "C" int rn_crop_boxes(const double *im_objs, const int64_t *cam_idxs, int n, double b, double *crop_boxes,
                             float *rois, void *stream) {
    if (n <= 0 || !im_objs || !crop_boxes || (rois && !cam_idxs)) return RN_EINVAL;
    hipLaunchKernelGGL(crop_boxes_kernel, dim3(rn_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, im_objs, cam_idxs, n, b,
                       crop_boxes, rois);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ---------------------------------------------------------------------------------------------------- roi_align
// torchvision's bilinear_interpolate, one channel plane
__device__ __forceinline__ float roi_bilinear(const float *__restrict__ d, int H, int W, float y, float x) {
    if (!(y >= -1.0f && y <= (float)H && x >= -1.0f && x <= (float)W)) return 0.f;   // also rejects NaN (torchvision's test lets it through)
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return ((w1 * d[(int64_t)y_low * W + x_low] + w2 * d[(int64_t)y_low * W + x_high]) + w3 * d[(int64_t)y_high * W + x_low]) +
           w4 * d[(int64_t)y_high * W + x_high];
}

// one lane per output pixel (all channels): consecutive lanes = consecutive pw, so the four taps of neighbouring lanes
// fall into the same or adjacent cache lines of the frame
__global__ __launch_bounds__(256) void roi_align_kernel(const float *__restrict__ frames, int N, int C, int H, int W,
                                                        const float *__restrict__ rois, int n, int ph_n, int pw_n,
                                                        float *__restrict__ out, int nhwc4) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (int64_t)ph_n * pw_n;
    if (t >= per * n) return;
    const int r = (int)(t / per);
    const int rem = (int)(t - (int64_t)r * per);
    const int ph = rem / pw_n, pw = rem - ph * pw_n;
    const float *roi = rois + (int64_t)r * 5;
    int b = (int)roi[0];
    b = b < 0 ? 0 : (b >= N ? N - 1 : b);                                      // torchvision would read out of bounds
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];           // spatial_scale 1, aligned = False
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bh = rh / (float)ph_n, bw = rw / (float)pw_n;
    // sampling_ratio = -1: ceil(roi size / output size) samples per bin and axis.  Capped at 64 (a 7 168-pixel crop at
    // cs = 112, larger than any frame): a non-finite or absurd box from a degenerate homography must not turn into an
    // unbounded loop on the GPU.  Within the cap the result is torchvision's.
    const int gh = (int)fminf(ceilf(rh / (float)ph_n), 64.f), gw = (int)fminf(ceilf(rw / (float)pw_n), 64.f);
    const float count = (float)(gh * gw > 1 ? gh * gw : 1);
    for (int c = 0; c < C; ++c) {
        const float *d = frames + ((int64_t)b * C + c) * H * W;
        float acc = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            const float y = (y1 + (float)ph * bh) + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = (x1 + (float)pw * bw) + ((float)ix + 0.5f) * bw / (float)gw;
                acc += roi_bilinear(d, H, W, y, x);
            }
        }
        const float v = acc / count;
        if (nhwc4) out[((int64_t)r * per + rem) * 4 + c] = v;
        else out[((int64_t)r * C + c) * per + rem] = v;
    }
    if (nhwc4)
        for (int c = C; c < 4; ++c) out[((int64_t)r * per + rem) * 4 + c] = 0.f;
}

extern "C" int rn_roi_align(const float *frames, int N, int C, int H, int W, const float *rois, int n, int out_h, int out_w,
                            float *out, int nhwc4, void *stream) {
    if (!frames || !rois || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || n <= 0 || out_h <= 0 || out_w <= 0) return RN_EINVAL;
    if (nhwc4 && C > 4) return RN_EINVAL;
    const int64_t total = (int64_t)n * out_h * out_w;
    hipLaunchKernelGGL(roi_align_kernel, dim3(rn_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, frames, N, C, H, W, rois,
                       n, out_h, out_w, out, nhwc4);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

// ---------------------------------------------------------------------------------------------------- candidate selection
#define CROP_MAX_A 4096          // anchors per crop the selection sorts in LDS (cs = 112 -> 2 394)
#define CROP_MAX_K 256           // cd_max

__device__ __forceinline__ unsigned crop_desc_key(float f) {                  // larger float -> smaller key
    unsigned u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}

// One workgroup per object.
__global__ __launch_bounds__(256) void crop_select_kernel(const float *__restrict__ reg_boxes, const float *__restrict__ cls,
                                                          const double *__restrict__ crop_boxes, const int64_t *__restrict__ cam_idxs,
                                                          const float *__restrict__ pre_loc, const double *__restrict__ H1,
                                                          const double *__restrict__ H2, const double *__restrict__ P1,
                                                          const double *__restrict__ P2, int n_cam, int A, int C, double cs,
                                                          int cd_max, float Wt, float *__restrict__ out_state,
                                                          int64_t *__restrict__ out_cls, float *__restrict__ out_conf) {
    __shared__ unsigned long long keys[CROP_MAX_A];
    __shared__ double s_score[CROP_MAX_K];
    __shared__ float s_state[CROP_MAX_K][6];
    const int o = blockIdx.x, tid = threadIdx.x;
    int npad = 64;
    while (npad < A) npad <<= 1;
    // confidence = max over classes (torch.max: first maximum), MC3D_crop_tracker.py:1193
    for (int a = tid; a < npad; a += 256) {
        unsigned long long k = ~0ull;
        if (a < A) {
            const float *p = cls + ((int64_t)o * A + a) * C;
            float best = p[0];
            for (int c = 1; c < C; ++c) best = p[c] > best ? p[c] : best;
            k = ((unsigned long long)crop_desc_key(best) << 32) | (unsigned)a;
        }
        keys[a] = k;
    }
    // torch.topk(confs, cd_max): decreasing confidence (equal confidences: lower anchor index first)
    for (int size = 2; size <= npad; size <<= 1)
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
            __syncthreads();
            for (int t = tid; t < (npad >> 1); t += 256) {
                const int lo = 2 * t - (t & (strd - 1)), hi = lo + strd;
                const bool up = (lo & size) == 0;
                const unsigned long long x = keys[lo], y = keys[hi];
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
        }
    __syncthreads();
    const int K = cd_max < A ? cd_max : A;
    int cam = (int)cam_idxs[o];
    cam = cam < 0 ? 0 : (cam >= n_cam ? n_cam - 1 : cam);
    const double *cb = crop_boxes + (int64_t)o * 4;
    const double scale = fmax(cb[2] - cb[0], cb[3] - cb[1]);                   // :955
    // prior footprint (select_best_box, :999-1006)
    float pst[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) pst[q] = pre_loc[(int64_t)o * 6 + q];
    const float4 pf = hg_footprint(pst);
    for (int k = tid; k < K; k += 256) {
        const int a = (int)(keys[k] & 0xffffffffu);
        const float *p = reg_boxes + ((int64_t)o * A + a) * 20;                // the 2D box (cols 16:20) is dropped, :953
        double2 pt[8];
        double bx[8], by[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {                                          // local_to_global, :958-963
            bx[q] = (double)p[2 * q] * scale / cs + cb[0];
            by[q] = (double)p[2 * q + 1] * scale / cs + cb[1];
            pt[q] = make_double2(bx[q], by[q]);
        }
        float st[6];
        hg_im_to_state_refined<double>(pt, bx, by, 5.0f, true, H1, H2, P1, P2, cam, st);   // :1213-1219, heights = "other"
        const float4 f = hg_footprint(st);
        // md_iou on .double() footprints, :1012
        const double area_a = ((double)f.z - (double)f.x) * ((double)f.w - (double)f.y);
        const double area_b = ((double)pf.z - (double)pf.x) * ((double)pf.w - (double)pf.y);
        const double minx = fmax((double)f.x, (double)pf.x), maxx = fmin((double)f.z, (double)pf.z);
        const double miny = fmax((double)f.y, (double)pf.y), maxy = fmin((double)f.w, (double)pf.w);
        const double inter = fmax(0.0, maxx - minx) * fmax(0.0, maxy - miny);
        const double iou = inter / ((area_a + area_b) - inter);
        // conf of this anchor again (the key holds only its order)
        const float *pc = cls + ((int64_t)o * A + a) * C;
        float conf = pc[0];
        for (int c = 1; c < C; ++c) conf = pc[c] > conf ? pc[c] : conf;
        s_score[k] = (1.0 - (double)Wt) * iou + (double)(Wt * conf);           // (1-W)*ious [f64] + W*confs [f32], :1015
#pragma unroll
        for (int q = 0; q < 6; ++q) s_state[k][q] = st[q];
    }
    __syncthreads();
    if (tid == 0) {                                                            // torch.argmax: first maximum, :1017
        int best = 0;
        for (int k = 1; k < K; ++k)
            if (s_score[k] > s_score[best]) best = k;                          // a NaN score never wins (torch would pick it)
        const int a = (int)(keys[best] & 0xffffffffu);
        const float *pc = cls + ((int64_t)o * A + a) * C;
        float conf = pc[0];
        int ci = 0;
        for (int c = 1; c < C; ++c)
            if (pc[c] > conf) { conf = pc[c]; ci = c; }
#pragma unroll
        for (int q = 0; q < 6; ++q) out_state[(int64_t)o * 6 + q] = s_state[best][q];
        out_cls[o] = ci;
        out_conf[o] = conf;
    }
}

extern "C" int rn_crop_select(const float *reg_boxes, const float *cls, const double *crop_boxes, const int64_t *cam_idxs,
                              const float *pre_loc, const double *H1, const double *H2, const double *P1, const double *P2,
                              int n_cam, int n, int A, int C, double cs, int cd_max, float W, float *out_state,
                              int64_t *out_cls, float *out_conf, void *stream) {
    if (n <= 0 || A <= 0 || C <= 0 || n_cam <= 0 || cd_max <= 0 || !H1 || !P1 || cs <= 0.0) return RN_EINVAL;
    if (A > CROP_MAX_A || cd_max > CROP_MAX_K) return RN_ETOOMANY;
    hipLaunchKernelGGL(crop_select_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, reg_boxes, cls, crop_boxes, cam_idxs,
                       pre_loc, H1, H2, P1, P2, n_cam, A, C, cs, cd_max, W, out_state, out_cls, out_conf);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
