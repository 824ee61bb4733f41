// Device functions of the I-24 homography shared by homography.hip (the standalone transforms) and
// tracker_post.hip (the fused detection parser).  Arithmetic and operation order follow homography.py; see the
// line references on each statement.
#pragma once
#include "common.h"

__device__ __forceinline__ void state_corners(const float *__restrict__ s, float x[8], float y[8], float z[8]) {
    const float xr = s[0], yc = s[1], l = s[2], w = s[3], h = s[4], dir = s[5];
    const float xf = xr + dir * l;                                              // homography.py:310
    const float half = dir * w / 2.0f;                                          // homography.py:314-315
    const float ylo = yc - half, yhi = yc + half;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = k & 3;
        x[k] = (q < 2) ? xf : xr;                                               // points {0,1,4,5} front
        y[k] = (k & 1) ? yhi : ylo;
        z[k] = (k >= 4) ? -h : 0.f;                                             // homography.py:318
    }
}

template <typename T>
__device__ __forceinline__ void corners_to_state(const T x[8], const T y[8], const T z[8], float *__restrict__ o) {
    const T fx = x[0] + x[1], rx = x[2] + x[3];
    o[0] = (float)(rx / (T)2.0);                                                // homography.py:286
    o[1] = (float)((((y[0] + y[1]) + y[2]) + y[3]) / (T)4.0);                   // homography.py:289
    const T dl = (fx - rx) / (T)2.0;
    o[2] = (float)(dl < 0 ? -dl : dl);                                          // homography.py:292
    const T dw = ((y[0] + y[2]) - (y[1] + y[3])) / (T)2.0;
    o[3] = (float)(dw < 0 ? -dw : dw);                                          // homography.py:295
    T hs = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const T e = z[k] - z[k + 4]; hs += (e < 0 ? -e : e); }
    o[4] = (float)(hs / (T)4.0);                                                // homography.py:298
    o[5] = (float)((dl > 0) - (dl < 0));                                        // homography.py:301
}

// im = P [x y z 1]^T, perspective divide (homography.py:438-476); P2 != nullptr is the Homography_Wrapper switch on the
// corner-0 space y (homography.py:849-856).  P / P2: per-camera 3x4 fp64 row-major, m = camera index.
__device__ __forceinline__ void hg_project_to_im(const float x[8], const float y[8], const float z[8],
                                                 const double *__restrict__ P, const double *__restrict__ P2, int m,
                                                 double2 out[8]) {
    const double *M = ((P2 != nullptr && y[0] > 60.0f) ? P2 : P) + (int64_t)m * 12;   // homography.py:854
    double pm[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) pm[k] = M[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double X = x[k], Y = y[k], Z = z[k];
        const double u = ((pm[0] * X + pm[1] * Y) + pm[2] * Z) + pm[3];
        const double v = ((pm[4] * X + pm[5] * Y) + pm[6] * Z) + pm[7];
        const double w = ((pm[8] * X + pm[9] * Y) + pm[10] * Z) + pm[11];
        out[k] = make_double2(u / w, v / w);                                    // homography.py:468-469
    }
}

// space = H [u v 1]^T divided, z = 0 for corners 0-3 and the height for 4-7 (homography.py:388-435); H2 != nullptr is
// the wrapper switch on hg1's corner-0 y (homography.py:840-847).  H / H2: per-camera 3x3 fp64 row-major.
__device__ __forceinline__ void hg_project_from_im(const double2 pt[8], double hgt, const double *__restrict__ H,
                                                   const double *__restrict__ H2, int m, double x[8], double y[8],
                                                   double z[8]) {
    auto project = [&](const double *M) {
        double hm[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) hm[k] = M[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double u = (hm[0] * pt[k].x + hm[1] * pt[k].y) + hm[2];
            const double v = (hm[3] * pt[k].x + hm[4] * pt[k].y) + hm[5];
            const double w = (hm[6] * pt[k].x + hm[7] * pt[k].y) + hm[8];
            x[k] = u / w;                                                       // homography.py:416-417
            y[k] = v / w;
            z[k] = k >= 4 ? hgt : 0.0;                                          // homography.py:426-428
        }
    };
    project(H + (int64_t)m * 9);
    if (H2 != nullptr && y[0] > 60.0) project(H2 + (int64_t)m * 9);              // homography.py:845-846
}

// torch.mean over 4 rows of one coordinate: rows are accumulated in order, then divided.
template <typename T>
__device__ __forceinline__ T hg_mean4(T a, T b, T c, T d) { return (((a + b) + c) + d) / (T)4; }
__device__ __forceinline__ float hg_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double hg_sqrt(double v) { return sqrt(v); }

// Image corners -> state through the camera's homographies, with the tracker's optional height refinement
// (MC3D_crop_tracker.py:366-370 and 1216-1219): state -> image through P (the reprojection), height_from_template
// (homography.py:519-551) of the reprojection (float64) against the detection itself, image -> state again.
// TB is the dtype in which height_from_template sees the detection: float32 in parse_detections (the detector's
// boxes), float64 in the crop path (local_to_global promotes them).  h0 = guess_heights value (float32).
template <typename TB>
__device__ __forceinline__ void hg_im_to_state_refined(const double2 pt[8], const TB bx[8], const TB by[8], float h0,
                                                       bool refine, const double *__restrict__ H1,
                                                       const double *__restrict__ H2, const double *__restrict__ P1,
                                                       const double *__restrict__ P2, int cam, float st[6]) {
    double x[8], y[8], z[8];
    hg_project_from_im(pt, (double)h0, H1, H2, cam, x, y, z);
    corners_to_state<double>(x, y, z, st);
    if (!refine) return;
    float fx[8], fy[8], fz[8];
    state_corners(st, fx, fy, fz);
    double2 rp[8];
    hg_project_to_im(fx, fy, fz, P1, P2, cam, rp);                           // repro_boxes = hg.state_to_im(boxes)
    const double ttx = hg_mean4(rp[4].x, rp[5].x, rp[6].x, rp[7].x), tty = hg_mean4(rp[4].y, rp[5].y, rp[6].y, rp[7].y);
    const double tbx = hg_mean4(rp[0].x, rp[1].x, rp[2].x, rp[3].x), tby = hg_mean4(rp[0].y, rp[1].y, rp[2].y, rp[3].y);
    const double dtx = ttx - tbx, dty = tty - tby;
    const double t_h = sqrt(dtx * dtx) + sqrt(dty * dty);
    const double ratio = t_h / (double)h0;
    const TB btx = hg_mean4(bx[4], bx[5], bx[6], bx[7]), bty = hg_mean4(by[4], by[5], by[6], by[7]);
    const TB bbx = hg_mean4(bx[0], bx[1], bx[2], bx[3]), bby = hg_mean4(by[0], by[1], by[2], by[3]);
    const TB dbx = btx - bbx, dby = bty - bby;
    const TB b_h = hg_sqrt(dbx * dbx) + hg_sqrt(dby * dby);
    const double h_ref = (double)b_h / ratio;
    hg_project_from_im(pt, h_ref, H1, H2, cam, x, y, z);
    corners_to_state<double>(x, y, z, st);
}

// road-plane footprint of a state: min / max of the four bottom corners, fp32 (MC3D_crop_tracker.py:626-633, 992-996)
__device__ __forceinline__ float4 hg_footprint(const float st[6]) {
    float sx[8], sy[8], sz[8];
    state_corners(st, sx, sy, sz);
    return make_float4(fminf(fminf(sx[0], sx[1]), fminf(sx[2], sx[3])), fminf(fminf(sy[0], sy[1]), fminf(sy[2], sy[3])),
                       fmaxf(fmaxf(sx[0], sx[1]), fmaxf(sx[2], sx[3])), fmaxf(fmaxf(sy[0], sy[1]), fmaxf(sy[2], sy[3])));
}
