// Batched Kalman filter of the tracker, one lane per object.
//
// Replaces the tensor algebra of Torch_KF.view / predict / update (util_track/kf.py:264-403): a handful of bmm / repeat /
// inverse calls per frame on [n,6,6] and [m,5,5] stacks (the reference keeps the filter on the CPU for that reason,
// MC3D_crop_tracker.py:103).  State per object: X[6] = (x, y, l, w, h, v), P[6][6], direction D, time T (float64).
// dtype quirks of the reference kept (oracle/kf.py): F[0][5] = D*dt; with a per-object dt tensor the model noise
// Q*dt/dt_default is formed in float64 and P rounded to float32, with a scalar dt everything is float32; the innovation
// is formed in float64 and rounded; S (5x5) is inverted in float32 (Gauss-Jordan with partial pivoting here, LAPACK
// getrf/getri there: agreement to round-off).  Latency-bound by construction: a few hundred objects.
#include <math.h>

#include "common.h"

#define KS 6   // state size
#define KM 5   // measurement size

__device__ __forceinline__ float kf_f05(float D, double dt, int dt_is_tensor) {
    return dt_is_tensor ? (float)((double)D * dt) : D * (float)dt;              // kf.py:278 / 310
}

__global__ __launch_bounds__(128) void kf_view_kernel(const float *__restrict__ X, const float *__restrict__ D,
                                                      const float *__restrict__ F, const double *__restrict__ dt,
                                                      int dt_is_tensor, int with_direction, float *__restrict__ out, int n) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    float x[KS], xp[KS];
#pragma unroll
    for (int a = 0; a < KS; ++a) x[a] = X[i * KS + a];
    if (dt != nullptr) {
        const float f05 = kf_f05(D[i], dt_is_tensor ? dt[i] : dt[0], dt_is_tensor);
#pragma unroll
        for (int a = 0; a < KS; ++a) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < KS; ++b) s += ((a == 0 && b == 5) ? f05 : F[a * KS + b]) * x[b];
            xp[a] = s;
        }
    } else {
#pragma unroll
        for (int a = 0; a < KS; ++a) xp[a] = x[a];
    }
    if (with_direction) {                                                       // cat(states[:, :-1], D, states[:, -1:]), kf.py:287
        float *o = out + (int64_t)i * (KS + 1);
#pragma unroll
        for (int a = 0; a < KS - 1; ++a) o[a] = xp[a];
        o[KS - 1] = D[i];
        o[KS] = xp[KS - 1];
    } else {
#pragma unroll
        for (int a = 0; a < KS; ++a) out[(int64_t)i * KS + a] = xp[a];
    }
}

__global__ __launch_bounds__(128) void kf_predict_kernel(float *__restrict__ X, float *__restrict__ P, const float *__restrict__ D,
                                                         double *__restrict__ T, const float *__restrict__ F,
                                                         const float *__restrict__ Q, const double *__restrict__ dt,
                                                         int dt_is_tensor, double dt_default, int n) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    const double dti = dt_is_tensor ? dt[i] : dt[0];
    float Fr[KS][KS], x[KS], p[KS][KS], fp[KS][KS];
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int b = 0; b < KS; ++b) Fr[a][b] = F[a * KS + b];
    Fr[0][5] = kf_f05(D[i], dti, dt_is_tensor);
#pragma unroll
    for (int a = 0; a < KS; ++a) x[a] = X[i * KS + a];
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int b = 0; b < KS; ++b) p[a][b] = P[(int64_t)i * KS * KS + a * KS + b];
#pragma unroll
    for (int a = 0; a < KS; ++a) {                                               // X = F_rep X, kf.py:311
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < KS; ++b) s += Fr[a][b] * x[b];
        X[i * KS + a] = s;
    }
#pragma unroll
    for (int a = 0; a < KS; ++a)                                                 // step1 = F P
#pragma unroll
        for (int b = 0; b < KS; ++b) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KS; ++c) s += Fr[a][c] * p[c][b];
            fp[a][b] = s;
        }
#pragma unroll
    for (int a = 0; a < KS; ++a)                                                 // step3 = step1 F^T, + Q scaled, kf.py:316-326
#pragma unroll
        for (int b = 0; b < KS; ++b) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KS; ++c) s += fp[a][c] * Fr[b][c];
            const float q = Q[a * KS + b];
            float r;
            if (dt_is_tensor) r = (float)((double)s + ((double)q * dti) / dt_default);
            else r = s + (q * (float)dti) / (float)dt_default;
            P[(int64_t)i * KS * KS + a * KS + b] = r;
        }
    T[i] += dti;                                                                 // kf.py:329
}

__global__ __launch_bounds__(64) void kf_update_kernel(float *__restrict__ X, float *__restrict__ P, const int32_t *__restrict__ rows,
                                                       const double *__restrict__ z, const float *__restrict__ H,
                                                       const float *__restrict__ R, const float *__restrict__ muR, int m) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= m) return;
    const int r = rows[k];
    float h[KM][KS], x[KS], p[KS][KS];
#pragma unroll
    for (int a = 0; a < KM; ++a)
#pragma unroll
        for (int b = 0; b < KS; ++b) h[a][b] = H[a * KS + b];
#pragma unroll
    for (int a = 0; a < KS; ++a) x[a] = X[r * KS + a];
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int b = 0; b < KS; ++b) p[a][b] = P[(int64_t)r * KS * KS + a * KS + b];
    float y[KM];
#pragma unroll
    for (int a = 0; a < KM; ++a) {                                               // y = z + mu_R - X H^T in float64, kf.py:373-377
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < KS; ++b) s += x[b] * h[a][b];
        y[a] = (float)((z[(int64_t)k * KM + a] + (double)muR[a]) - (double)s);
    }
    float hp[KM][KS], S[KM][KM], pht[KS][KM];
#pragma unroll
    for (int a = 0; a < KM; ++a)
#pragma unroll
        for (int b = 0; b < KS; ++b) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KS; ++c) s += h[a][c] * p[c][b];
            hp[a][b] = s;
        }
#pragma unroll
    for (int a = 0; a < KM; ++a)
#pragma unroll
        for (int b = 0; b < KM; ++b) {                                           // S = H P H^T + R, kf.py:382-385
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KS; ++c) s += hp[a][c] * h[b][c];
            S[a][b] = s + R[a * KM + b];
        }
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int b = 0; b < KM; ++b) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KS; ++c) s += p[a][c] * h[b][c];
            pht[a][b] = s;
        }
    // S^-1 by Gauss-Jordan with partial pivoting; every index is a compile-time constant (row swaps are selects)
    float inv[KM][KM];
#pragma unroll
    for (int a = 0; a < KM; ++a)
#pragma unroll
        for (int b = 0; b < KM; ++b) inv[a][b] = a == b ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < KM; ++c) {
        int piv = c;
        float best = fabsf(S[c][c]);
#pragma unroll
        for (int a = c + 1; a < KM; ++a) {
            const float v = fabsf(S[a][c]);
            if (v > best) { best = v; piv = a; }
        }
#pragma unroll
        for (int a = c + 1; a < KM; ++a) {
            const bool sw = piv == a;
#pragma unroll
            for (int b = 0; b < KM; ++b) {
                const float s0 = S[c][b], s1 = S[a][b], i0 = inv[c][b], i1 = inv[a][b];
                S[c][b] = sw ? s1 : s0; S[a][b] = sw ? s0 : s1;
                inv[c][b] = sw ? i1 : i0; inv[a][b] = sw ? i0 : i1;
            }
        }
        const float d = 1.0f / S[c][c];
#pragma unroll
        for (int b = 0; b < KM; ++b) { S[c][b] *= d; inv[c][b] *= d; }
#pragma unroll
        for (int a = 0; a < KM; ++a) {
            if (a == c) continue;
            const float f = S[a][c];
#pragma unroll
            for (int b = 0; b < KM; ++b) { S[a][b] -= f * S[c][b]; inv[a][b] -= f * inv[c][b]; }
        }
    }
    float K[KS][KM];
#pragma unroll
    for (int a = 0; a < KS; ++a)
#pragma unroll
        for (int b = 0; b < KM; ++b) {                                           // K = P H^T S^-1, kf.py:388-389
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KM; ++c) s += pht[a][c] * inv[c][b];
            K[a][b] = s;
        }
#pragma unroll
    for (int a = 0; a < KS; ++a) {                                               // X += K y, kf.py:393-395
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < KM; ++b) s += K[a][b] * y[b];
        X[r * KS + a] = x[a] + s;
    }
#pragma unroll
    for (int a = 0; a < KS; ++a) {                                               // P = (I - K H) P, kf.py:398-400
        float ikh[KS];
#pragma unroll
        for (int b = 0; b < KS; ++b) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KM; ++c) s += K[a][c] * h[c][b];
            ikh[b] = (a == b ? 1.f : 0.f) - s;
        }
#pragma unroll
        for (int b = 0; b < KS; ++b) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < KS; ++c) s += ikh[c] * p[c][b];
            P[(int64_t)r * KS * KS + a * KS + b] = s;
        }
    }
}

extern "C" int rn_kf_view(const float *X, const float *D, const float *F, const double *dt, int dt_is_tensor, int with_direction,
                          float *out, int n, void *stream) {
    if (n <= 0 || !X || !D || !F || !out) return RN_EINVAL;
    hipLaunchKernelGGL(kf_view_kernel, dim3(rn_blocks(n, 128)), dim3(128), 0, (hipStream_t)stream, X, D, F, dt, dt_is_tensor,
                       with_direction, out, n);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_kf_predict(float *X, float *P, const float *D, double *T, const float *F, const float *Q, const double *dt,
                             int dt_is_tensor, double dt_default, int n, void *stream) {
    if (n <= 0 || !X || !P || !D || !T || !F || !Q || !dt || dt_default == 0.0) return RN_EINVAL;
    hipLaunchKernelGGL(kf_predict_kernel, dim3(rn_blocks(n, 128)), dim3(128), 0, (hipStream_t)stream, X, P, D, T, F, Q, dt,
                       dt_is_tensor, dt_default, n);
    RN_LAUNCH_CHECK();
    return RN_OK;
}

extern "C" int rn_kf_update(float *X, float *P, const int32_t *rows, const double *z, const float *H, const float *R,
                            const float *mu_R, int m, void *stream) {
    if (m <= 0 || !X || !P || !rows || !z || !H || !R || !mu_R) return RN_EINVAL;
    hipLaunchKernelGGL(kf_update_kernel, dim3(rn_blocks(m, 64)), dim3(64), 0, (hipStream_t)stream, X, P, rows, z, H, R, mu_R, m);
    RN_LAUNCH_CHECK();
    return RN_OK;
}
