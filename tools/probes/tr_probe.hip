// Probe for gfx950's transposing LDS read as the bf16 weight-gradient kernel uses it (conv_bf16.hip):
// image = [32 pixels][128 channels] of 16-bit values, 256-byte rows, 16-byte chunks XOR-permuted per row
// (cdna_hip_programming.md T10, image (b)); operand(lane) = 8 consecutive pixels of one channel.
// Prints the number of mismatching elements for every (channel sub-tile, pixel half) and the bank-conflict-relevant
// address pattern is left to the PMC run.   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define ROWB 256
__device__ __host__ inline int fx(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ inline unsigned off(int row, int ch) { return ROWB * row + 16 * (ch ^ fx(row)); }

__global__ void probe(int *bad, short *dump) {
    __shared__ __attribute__((aligned(16))) short lds[32 * 128];
    for (int i = threadIdx.x; i < 32 * 128; i += 64) {          // logical element (pixel p, channel c) = p * 256 + c
        const int p = i / 128, c = i % 128;
        const unsigned o = off(p, c / 8) + 2 * (c % 8);
        lds[o / 2] = (short)(p * 256 + c);
    }
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    int nbad = 0;
    for (int sub = 0; sub < 4; ++sub)                            // 32-channel sub-tile of the 128
        for (int kh = 0; kh < 2; ++kh) {                         // pixels 16*kh .. 16*kh+15
            short got[8];
            for (int rd = 0; rd < 2; ++rd) {                     // two reads: pixels +0..3 and +4..7 of this lane's k-group
                const int r0 = 16 * kh + 8 * (g >> 1) + 4 * rd;  // block's first pixel
                const int c0 = (32 * sub + 16 * (g & 1)) / 8;    // block's first chunk (16 channels = 2 chunks)
                const unsigned a = off(r0 + q, c0 + (pp >> 1)) + 8 * (pp & 1);
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4 *)((__attribute__((address_space(3))) char *)lds + a));
                for (int e = 0; e < 4; ++e) got[4 * rd + e] = v[e];
            }
            const int ch = 32 * sub + (lane & 31);               // MFMA operand: row i = lane & 31, k = 8 * (lane >> 5) + e
            for (int e = 0; e < 8; ++e) {
                const int p = 16 * kh + 8 * (lane >> 5) + e;
                if (got[e] != (short)(p * 256 + ch)) ++nbad;
                if (sub == 1 && kh == 1) dump[lane * 8 + e] = got[e];
            }
        }
    bad[lane] = nbad;
}

int main() {
    int *bad; short *dump;
    hipMalloc(&bad, 64 * 4); hipMalloc(&dump, 64 * 8 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, bad, dump);
    int h[64]; short hd[512];
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hd, dump, sizeof(hd), hipMemcpyDeviceToHost);
    int tot = 0; for (int i = 0; i < 64; ++i) tot += h[i];
    printf("mismatching elements: %d of %d\n", tot, 64 * 8 * 8);
    for (int l = 0; l < 64; l += 13) { printf("lane %2d sub 1 kh 1:", l); for (int e = 0; e < 8; ++e) printf(" (%d,%d)", hd[l * 8 + e] / 256, hd[l * 8 + e] % 256); printf("\n"); }
    return tot != 0;
}
