// What does a raw buffer load return out of range on gfx950?  Settles how conv_wgrad / conv_igemm may use the
// hardware range check instead of masks:   hipcc --offload-arch=gfx950 -O3 -o buffer_probe buffer_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void probe(const float *p, unsigned num_records, const unsigned *voff, const unsigned *soff, float4 *out, int n) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, num_records, 0x00020000);
    for (int i = 0; i < n; ++i) {
        const unsigned so = soff[i];                                  // uniform
        v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, voff[i], __builtin_amdgcn_readfirstlane(so), 0);
        if (threadIdx.x == 0) out[i] = *reinterpret_cast<float4 *>(&v);
    }
}

int main() {
    const int N = 1024;                                               // floats in the allocation
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = 1000.f + i;
    float *d; hipMalloc(&d, N * 4); hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    const unsigned nr = 512 * 4;                                      // descriptor covers the first 512 floats only
    struct { unsigned v, s; const char *what; } cases[] = {
        {0, 0, "in range"},
        {nr - 16, 0, "last full 16 B"},
        {nr - 8, 0, "straddles the end (8 B in, 8 B out)"},
        {nr, 0, "voffset == num_records"},
        {nr + 160, 0, "voffset > num_records (memory is mapped)"},
        {0xFFFFFFFFu, 0, "voffset = -1"},
        {0x80000000u + 64, 0, "voffset = 2^31 + 64"},
        {0, nr, "soffset == num_records, voffset 0"},
        {nr - 16, 16, "voffset in range, voffset+soffset out"},
        {64, 128, "both in range"},
        {0xFFFFFFF0u, 32, "voffset -16, soffset 32 (sum wraps into range)"},
    };
    const int n = sizeof(cases) / sizeof(cases[0]);
    std::vector<unsigned> v(n), s(n);
    for (int i = 0; i < n; ++i) { v[i] = cases[i].v; s[i] = cases[i].s; }
    unsigned *dv, *ds; float4 *dout;
    hipMalloc(&dv, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dout, n * 16);
    hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, nr, dv, ds, dout, n);
    std::vector<float4> o(n);
    hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
    printf("buffer of %u bytes (floats 1000..1511); memory behind it holds 1512..2023\n", nr);
    for (int i = 0; i < n; ++i)
        printf("%-52s voff %10u soff %5u -> %7.0f %7.0f %7.0f %7.0f\n", cases[i].what, cases[i].v, cases[i].s, o[i].x, o[i].y, o[i].z, o[i].w);
    return 0;
}
