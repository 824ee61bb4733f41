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
