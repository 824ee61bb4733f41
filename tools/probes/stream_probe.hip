// Floor for short streaming-read kernels on MI355X: how fast can ~100 MB be read and reduced, by launch shape?
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe
// mode 0: one-shot tiles (each 256-thread workgroup reads TILE_F4*16 B, 8 float4 per lane in flight) -- the loss kernel's shape
// mode 1: persistent grid-stride (G workgroups loop over the buffer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void oneshot(const float4 *__restrict__ x, long n4, float *out) {
    const long base = (long)blockIdx.x * 2048;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { long i = base + k * 256 + threadIdx.x; v[k] = x[i < n4 ? i : n4 - 1]; }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    if (s == 12345.678f) out[0] = s;          // keep the loads alive without a reduction tail
}
__global__ __launch_bounds__(256) void oneshot4(const float4 *__restrict__ x, long n4, float *out) {   // 16 KB per WG
    const long base = (long)blockIdx.x * 1024;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { long i = base + k * 256 + threadIdx.x; v[k] = x[i < n4 ? i : n4 - 1]; }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void persistent(const float4 *__restrict__ x, long n4, float *out) {
    float s = 0.f;
    const long stride = (long)gridDim.x * 2048;
    for (long base = (long)blockIdx.x * 2048; base < n4; base += stride) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { long i = base + k * 256 + threadIdx.x; v[k] = x[i < n4 ? i : n4 - 1]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void empty_kernel(float *out) { if (threadIdx.x == 9999) out[0] = 1.f; }

int main() {
    const long sizes[] = {100L << 20, 400L << 20, 1600L << 20};
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {   // launch floor: empty kernels back to back
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, out);
        hipDeviceSynchronize(); hipEventRecord(e0);
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, out);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel back-to-back: %.2f us per launch\n", ms * 10.f);
        for (int g : {3048, 30480}) {
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, 0, out);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("empty kernel, %d workgroups of 256: %.2f us per launch\n", g, ms * 20.f);
        }
    }
    for (long bytes : sizes) {
        const int NBUF = bytes <= (400L << 20) ? 6 : 2;       // rotate buffers: defeat the 256 MB Infinity Cache
        std::vector<float4 *> bufs(NBUF);
        for (auto &b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 0, bytes); }
        const long n4 = bytes / 16;
        auto run = [&](const char *name, int mode, int grid) {
            const int iters = 30;
            for (int i = 0; i < 3; ++i) {
                if (mode == 0) hipLaunchKernelGGL(oneshot, dim3((n4 + 2047) / 2048), dim3(256), 0, 0, bufs[i % NBUF], n4, out);
                else if (mode == 2) hipLaunchKernelGGL(oneshot4, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, bufs[i % NBUF], n4, out);
                else hipLaunchKernelGGL(persistent, dim3(grid), dim3(256), 0, 0, bufs[i % NBUF], n4, out);
            }
            hipDeviceSynchronize(); hipEventRecord(e0);
            for (int i = 0; i < iters; ++i) {
                if (mode == 0) hipLaunchKernelGGL(oneshot, dim3((n4 + 2047) / 2048), dim3(256), 0, 0, bufs[i % NBUF], n4, out);
                else if (mode == 2) hipLaunchKernelGGL(oneshot4, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, bufs[i % NBUF], n4, out);
                else hipLaunchKernelGGL(persistent, dim3(grid), dim3(256), 0, 0, bufs[i % NBUF], n4, out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / iters;
            printf("%5ld MB  %-28s %8.1f us  %7.1f GB/s\n", bytes >> 20, name, us, bytes / us / 1e3);
        };
        run("oneshot 32KB/wg", 0, 0);
        run("oneshot 16KB/wg", 2, 0);
        run("persistent 1024 wg", 1, 1024);
        run("persistent 2048 wg", 1, 2048);
        run("persistent 4096 wg", 1, 4096);
        for (auto &b : bufs) hipFree(b);
    }
    return 0;
}
