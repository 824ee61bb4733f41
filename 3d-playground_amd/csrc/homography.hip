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
#include "homography_dev.h"

__global__ void state_to_space_kernel(const float *__restrict__ state, float *__restrict__ space, int64_t d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    float x[8], y[8], z[8];
    state_corners(state + i * 6, x, y, z);
    float *o = space + i * 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) { o[3 * k] = x[k]; o[3 * k + 1] = y[k]; o[3 * k + 2] = z[k]; }
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
    double2 *o = reinterpret_cast<double2 *>(im + i * 16);
    hg_project_to_im(x, y, z, P, P2, m, o);
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
    hg_project_from_im(pt, (double)heights[i], H, H2, m, x, y, z);
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
