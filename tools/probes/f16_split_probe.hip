// Probe for RN_FP32_SPLIT3 (csrc/mfma_split.h, second half): x*s = hi + lo with hi = f16(x*s), lo = f16(x*s - hi), and
// a*b ~= ah*bh + ah*bl + al*bh on v_mfma_f32_16x16x32_f16.  Answers, on the hardware:
//   1. how exact hi + lo is, element by element, over 40 binades below the tensor's maximum (relative error in the full-precision
//      window, absolute error below it);
//   2. whether the matrix core keeps fp16 SUBNORMAL operands (lo of a small element is one): a product of a subnormal and 1.0;
//   3. the error of a K = 32 dot product formed from the three term products against fp64, beside the fp32 FMA chain's.
//   hipcc --offload-arch=gfx950 -O3 -I ../../3d-playground_amd/csrc f16_split_probe.hip -o f16_split_probe && ./f16_split_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfma_split.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void split_kernel(const float *x, float *hi, float *lo, int n, float s) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    unsigned h, l;
    split_pair_h(x[i], x[i + 1], s, h, l);
    const f16x2 hp = __builtin_bit_cast(f16x2, h), lp = __builtin_bit_cast(f16x2, l);
    hi[i] = (float)hp[0]; hi[i + 1] = (float)hp[1];
    lo[i] = (float)lp[0]; lo[i + 1] = (float)lp[1];
}

// one wave: D = A * B for one 16 x 16 tile, K = 32; A[i][k], B[k][j] given as fp32, split with the scales sa, sb
__global__ void mfma_kernel(const float *A, const float *B, float *D, float sa, float sb, int terms) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    float av[8], bv[8];
    for (int j = 0; j < 8; ++j) { av[j] = A[r * 32 + 8 * g + j]; bv[j] = B[(8 * g + j) * 16 + r]; }
    const SplitH8 a = split8h(av, sa), b = split8h(bv, sb);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (terms == 3) RN_SPLITH_MFMA16(acc, a, b);
    else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.l, b.h, acc, 0, 0, 0);     // only lo(A) * hi(B): the subnormal test
    for (int e = 0; e < 4; ++e) D[(4 * g + e) * 16 + r] = acc[e];
}

int main() {
    const int n = 1 << 20;
    float *x = (float *)malloc(n * 4), *hi = (float *)malloc(n * 4), *lo = (float *)malloc(n * 4);
    srand(7);
    // magnitudes log-uniform over 40 binades below 1.0 (the "tensor maximum"), random mantissas and signs
    for (int i = 0; i < n; ++i) {
        const double e = -40.0 * (rand() / (double)RAND_MAX), m = 1.0 + rand() / (double)RAND_MAX;
        x[i] = (float)((rand() & 1 ? -1.0 : 1.0) * m * 0.5 * exp2(e));
    }
    x[0] = 0.999999f;
    const float s = 16384.f;                                  // amax < 1: 2^(14 - (-1))... amax in [2^-1, 1) -> scale 2^15; 2^14 keeps a binade of head-room here
    float *dx, *dh, *dl;
    hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    split_kernel<<<n / 512, 256>>>(dx, dh, dl, n, s);
    hipMemcpy(hi, dh, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(lo, dl, n * 4, hipMemcpyDeviceToHost);
    double worst_rel[8] = {0}, worst_abs = 0, sum2[8] = {0};
    long cnt[8] = {0};
    for (int i = 0; i < n; ++i) {
        const double t = (double)x[i] * s, err = fabs(t - ((double)hi[i] + (double)lo[i]));
        const int bucket = (int)fmin(7.0, floor(-log2(fabs((double)x[i]) + 1e-300) / 5.0));      // 5 binades per bucket
        const double rel = err / fabs(t);
        if (rel > worst_rel[bucket]) worst_rel[bucket] = rel;
        sum2[bucket] += rel * rel; cnt[bucket]++;
        if (bucket >= 4 && err > worst_abs) worst_abs = err;          // below the full-precision window: lo is a subnormal
    }
    printf("1. hi + lo against x * s (s = 2^14, tensor maximum ~1.0 -> 2^14), %d values over 40 binades:\n", n);
    for (int b = 0; b < 8; ++b)
        printf("   |x| in [2^-%d, 2^-%d): worst relative error 2^%.2f, rms 2^%.2f  (%ld values)\n", 5 * b + 5, 5 * b,
               log2(worst_rel[b] + 1e-300), 0.5 * log2(sum2[b] / (cnt[b] ? cnt[b] : 1) + 1e-300), cnt[b]);
    printf("   worst absolute error of the elements below 2^-20 of the maximum, in scaled units: 2^%.2f (half an fp16 subnormal quantum = 2^-25)\n", log2(worst_abs + 1e-300));

    // 2. subnormal operands
    float A[16 * 32] = {0}, B[32 * 16] = {0}, D[256];
    for (int i = 0; i < 12; ++i) A[i * 32] = 1.0f + ldexpf(1.0f, -12 - i);      // hi = 1.0, lo = 2^(-12 - i): subnormal from i = 3 on (1 + 2^-23 is fp32's last)
    for (int j = 0; j < 16; ++j) B[j] = 1.0f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dD, sizeof(D));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(dA, dB, dD, 1.0f, 1.0f, 1);
    hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    printf("2. lo(A) * hi(B) with lo = 2^(-12 - i), B = 1 (fp16 normals end at 2^-14, subnormals at 2^-24):\n   ");
    int kept = 0;
    for (int i = 0; i < 12; ++i) { printf("i=%d:%s ", i, D[i * 16] == ldexpf(1.0f, -12 - i) ? "exact" : (D[i * 16] == 0.f ? "ZERO" : "other")); kept += D[i * 16] != 0.f; }
    printf("\n   -> subnormal operands %s by v_mfma_f32_16x16x32_f16\n", kept == 12 ? "are KEPT" : "are FLUSHED");

    // 3. a K = 32 dot product: three term products against fp64, beside a plain fp32 fma chain
    double e3 = 0, ef = 0, ref2 = 0;
    for (int trial = 0; trial < 64; ++trial) {
        for (int i = 0; i < 16 * 32; ++i) { A[i] = (float)((rand() / (double)RAND_MAX - 0.5) * 8.0); B[i] = (float)((rand() / (double)RAND_MAX - 0.5) * 0.2); }
        hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
        mfma_kernel<<<1, 64>>>(dA, dB, dD, 4096.f, 131072.f, 3);              // amax 4 -> 2^14, amax 0.1 -> 2^13.7
        hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0; float f = 0.f;
                for (int k = 0; k < 32; ++k) { ref += (double)A[i * 32 + k] * B[k * 16 + j]; f = fmaf(A[i * 32 + k], B[k * 16 + j], f); }
                const double got = (double)D[i * 16 + j] / (4096.0 * 131072.0);
                e3 += (got - ref) * (got - ref); ef += ((double)f - ref) * ((double)f - ref); ref2 += ref * ref;
            }
    }
    printf("3. K = 32 dot products of uniform data: rms error / rms value: three fp16 term products %.3e, fp32 fma chain %.3e\n",
           sqrt(e3 / ref2), sqrt(ef / ref2));
    return 0;
}
