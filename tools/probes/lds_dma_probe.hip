// Direct-to-LDS buffer loads (16 bytes per lane) on gfx950: where do lanes land, and what do out-of-range lanes
// write?   hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float *p, unsigned num_records, float *out) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 777.f;     // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, num_records, 0x00020000);
    const int lane = threadIdx.x;
    // lane l reads 16 bytes at a scattered offset; lanes 8..15 are out of range (-1), lanes 16..23 beyond num_records
    unsigned voff = ((lane * 7) % 64) * 16;
    if (lane >= 8 && lane < 16) voff = 0xFFFFFFFFu;
    if (lane >= 16 && lane < 24) voff = num_records + lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds + 256, 16, voff, 0, 0, 0);     // destination: lds[256 ...]
    __builtin_amdgcn_s_waitcnt(0);        // vmcnt(0) lgkmcnt(0) expcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
    const int N = 4096;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *d, *o; hipMalloc(&d, N * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1024u, o);
    std::vector<float> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    printf("lds[252..255] (before the destination): %g %g %g %g\n", r[252], r[253], r[254], r[255]);
    for (int l = 0; l < 64; ++l) {
        const float *v = &r[256 + 4 * l];
        unsigned src = ((l * 7) % 64) * 4;
        printf("lane %2d -> lds[%3d..]: %5g %5g %5g %5g   (%s, source float %u)\n", l, 256 + 4 * l, v[0], v[1], v[2], v[3],
               (l >= 8 && l < 24) ? "OUT OF RANGE" : "in range", src);
    }
    printf("lds[512..515] (behind the destination): %g %g %g %g\n", r[512], r[513], r[514], r[515]);
    return 0;
}
