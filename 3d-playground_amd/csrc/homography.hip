// State <-> space <-> image transforms of the I-24 homography on device.
//
// Replaces Homography.i24_state_to_space / i24_space_to_state (homography.py:305-320 / 274-303),
// space_to_im / im_to_space (homography.py:438-476 / 388-435), their compositions state_to_im / im_to_state
// (homography.py:479-500) and the two-homography switch of Homography_Wrapper (homography.py:840-862).
// The reference runs these on the CPU with per-object matrices stacked by a Python list comprehension and a
// float64 bmm (3.0 / 3.9 ms for 3 600 boxes).
//
// dtype contract kept: state and state_to_space are fp32; projection arithmetic and image points are fp64
// (`.double()` at homography.py:401,453); im_to_state rounds fp64 results into an fp32 [d,6].
// One lane per object (8 points each); trivially latency-bound -- report microseconds, not GB/s.
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

__global__ void state_to_space_kernel(const float *__restrict__ state, float *__restrict__ space, int64_t d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    float x[8], y[8], z[8];
    state_corners(state + i * 6, x, y, z);
    float *o = space + i * 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) { o[3 * k] = x[k]; o[3 * k + 1] = y[k]; o[3 * k + 2] = z[k]; }
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

__global__ void space_to_state_kernel(const double *__restrict__ space, float *__restrict__ state, int64_t d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    double x[8], y[8], z[8];
    const double *p = space + i * 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = p[3 * k]; y[k] = p[3 * k + 1]; z[k] = p[3 * k + 2]; }
    corners_to_state<double>(x, y, z, state + i * 6);
}

// im = P [x y z 1]^T, perspective divide.  FROM_STATE builds the corners first (state_to_im); otherwise the
// fp32 corners are read from `space` (space_to_im).
template <bool FROM_STATE>
__global__ void to_im_kernel(const float *__restrict__ in, const double *__restrict__ P, const double *__restrict__ P2,
                             const int32_t *__restrict__ mat_index, double *__restrict__ im, int64_t d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    float x[8], y[8], z[8];
    if (FROM_STATE) {
        state_corners(in + i * 6, x, y, z);
    } else {
        const float *p = in + i * 24;
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = p[3 * k]; y[k] = p[3 * k + 1]; z[k] = p[3 * k + 2]; }
    }
    const int m = mat_index ? mat_index[i] : 0;
    const double *M = ((P2 != nullptr && y[0] > 60.0f) ? P2 : P) + (int64_t)m * 12;   // homography.py:854
    double pm[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) pm[k] = M[k];
    double2 *o = reinterpret_cast<double2 *>(im + i * 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double X = x[k], Y = y[k], Z = z[k];
        const double u = ((pm[0] * X + pm[1] * Y) + pm[2] * Z) + pm[3];
        const double v = ((pm[4] * X + pm[5] * Y) + pm[6] * Z) + pm[7];
        const double w = ((pm[8] * X + pm[9] * Y) + pm[10] * Z) + pm[11];
        o[k] = make_double2(u / w, v / w);                                      // homography.py:468-469
    }
}

// space = H [u v 1]^T divided, z = 0 for corners 0-3 and heights for 4-7.  TO_STATE folds i24_space_to_state in.
template <bool TO_STATE>
__global__ void from_im_kernel(const double *__restrict__ im, const float *__restrict__ heights,
                               const double *__restrict__ H, const double *__restrict__ H2,
                               const int32_t *__restrict__ mat_index, double *__restrict__ space,
                               float *__restrict__ state, int64_t d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    const int m = mat_index ? mat_index[i] : 0;
    const double2 *p = reinterpret_cast<const double2 *>(im + i * 16);
    double2 pt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pt[k] = p[k];
    double x[8], y[8], z[8];
    const double hgt = (double)heights[i];
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
    if (TO_STATE) {
        corners_to_state<double>(x, y, z, state + i * 6);
    } else {
        double *o = space + i * 24;
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[3 * k] = x[k]; o[3 * k + 1] = y[k]; o[3 * k + 2] = z[k]; }
    }
}

#define HG_LAUNCH(kernel, ...)                                                                           \
    do {                                                                                                 \
        if (d <= 0) return RN_EINVAL;                                                                    \
        hipLaunchKernelGGL(kernel, dim3(rn_blocks(d, 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        RN_LAUNCH_CHECK();                                                                               \
        return RN_OK;                                                                                    \
    } while (0)

extern "C" int rn_state_to_space(const float *state, float *space, int64_t d, void *stream) {
    HG_LAUNCH(state_to_space_kernel, state, space, d);
}
extern "C" int rn_space_to_state(const double *space, float *state, int64_t d, void *stream) {
    HG_LAUNCH(space_to_state_kernel, space, state, d);
}
extern "C" int rn_state_to_im(const float *state, const double *P, const double *P2, const int32_t *mat_index,
                              double *im, int64_t d, void *stream) {
    HG_LAUNCH(to_im_kernel<true>, state, P, P2, mat_index, im, d);
}
extern "C" int rn_space_to_im(const float *space, const double *P, const double *P2, const int32_t *mat_index,
                              double *im, int64_t d, void *stream) {
    HG_LAUNCH(to_im_kernel<false>, space, P, P2, mat_index, im, d);
}
extern "C" int rn_im_to_space(const double *im, const float *heights, const double *H, const double *H2,
                              const int32_t *mat_index, double *space, int64_t d, void *stream) {
    HG_LAUNCH(from_im_kernel<false>, im, heights, H, H2, mat_index, space, (float *)nullptr, d);
}
extern "C" int rn_im_to_state(const double *im, const float *heights, const double *H, const double *H2,
                              const int32_t *mat_index, float *state, int64_t d, void *stream) {
    HG_LAUNCH(from_im_kernel<true>, im, heights, H, H2, mat_index, (double *)nullptr, state, d);
}
