// What does the ACCESS PATTERN of the activation operand cost?  The fp16-split implicit GEMM (csrc/conv_igemm_mf16.hip) with its MFMAs and
// its stores knocked out (RN_MF16_KO=5) reads the Winograd stage's V tensor (792 576 rows of 1 KB) at 2.7 TB/s, where the streaming
// transforms reach 5-6.  This probe reads the same tensor with the kernel's skeleton -- 128-row tiles, four waves of 32 rows, K-steps
// separated by a workgroup barrier, DEPTH steps in flight -- and varies only how many bytes of a row one K-step takes (WC) and which
// lane takes them:
//   PAT 0  the kernel's layout: lane (r = lane & 15, g = lane >> 4) reads WC / 4 contiguous bytes at g * WC / 4 of row 16 * sm + r
//          (WC = 128: two b128 per 16-row block = what v_mfma_f32_16x16x32 needs of a 32-value K-step)
//   PAT 1  row-coalesced: WC / 16 consecutive lanes read one row's WC bytes (what a direct-to-LDS staging would issue)
// and the workgroups per CU (dynamic LDS as ballast).
//   hipcc --offload-arch=gfx950 -O3 a_pattern_probe.hip -o a_pattern_probe && ./a_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWB = 1024, BM = 128;

template <int WC, int DEPTH, int PAT>
__global__ __launch_bounds__(256) void probe_kernel(const float *__restrict__ x, float *__restrict__ out, int tiles, int wrap) {
    extern __shared__ char ballast[];
    constexpr int STEPS = ROWB / WC;
    constexpr int NL = WC / 64;                       // b128 loads per 16-row block and step (PAT 0) -- 2 * NL per lane and step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroup id -> tile: ids are dealt round-robin to the 8 XCDs; give every XCD a contiguous range of tiles (as xcd_remap does)
    const int per = (tiles + 7) / 8;
    int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= tiles) return;
    if (wrap > 0) tile %= wrap;          // L2-resident variant: every workgroup re-reads one of `wrap` tiles
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), (short)0, (int)((unsigned)tiles * BM * ROWB > 0x7fffffffu ? 0x7fffffff : tiles * BM * ROWB), 0x00020000);
    int base[2 * NL];
    if (PAT == 0) {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int i = 0; i < NL; ++i) base[sm * NL + i] = (tile * BM + 32 * wave + 16 * sm + r) * ROWB + g * (WC / 4) + 16 * i;
    } else if (PAT == 2) {                            // as PAT 0, but the four lanes of a row take 64 CONTIGUOUS bytes per instruction
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int sm = 0; sm < 2; ++sm)
#pragma unroll
            for (int i = 0; i < NL; ++i) base[sm * NL + i] = (tile * BM + 32 * wave + 16 * sm + r) * ROWB + g * 16 + 64 * i;
    } else {
        constexpr int LPR = WC / 16, RPI = 64 / LPR;   // lanes per row, rows per instruction; 32 rows per wave = 32 / RPI = 2 * NL instructions
#pragma unroll
        for (int i = 0; i < 2 * NL; ++i) base[i] = (tile * BM + 32 * wave + RPI * i + lane / LPR) * ROWB + (lane % LPR) * 16;
    }
    f32x4 buf[DEPTH][2 * NL];
    auto issue = [&](int s, int slot) {
#pragma unroll
        for (int i = 0; i < 2 * NL; ++i)
            buf[slot][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base[i], s * WC, 0));
    };
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) issue(s, s);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int slot = s % DEPTH;
#pragma unroll
        for (int i = 0; i < 2 * NL; ++i) acc += buf[slot][i][0] + buf[slot][i][1] + buf[slot][i][2] + buf[slot][i][3];
        if (s + DEPTH < STEPS) issue(s + DEPTH, slot);
        __syncthreads();
    }
    if (acc == 1.2345f) out[blockIdx.x] = acc + ballast[threadIdx.x];
}

template <int WC, int DEPTH, int PAT>
static void run(const float *x, float *out, int tiles, int wgs_per_cu, int wrap = 0) {
    const int lds = wgs_per_cu >= 5 ? 28 * 1024 : 160 * 1024 / wgs_per_cu - 2048;
    CK(hipFuncSetAttribute((const void *)probe_kernel<WC, DEPTH, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = (tiles + 7) / 8 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe_kernel<WC, DEPTH, PAT>), dim3(grid), dim3(256), lds, 0, x, out, tiles, wrap);
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((probe_kernel<WC, DEPTH, PAT>), dim3(grid), dim3(256), lds, 0, x, out, tiles, wrap);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= it;
    printf("  WC %4d B/row/step  depth %d  pattern %d  %d wg/CU%s: %.3f ms  %.2f TB/s\n", WC, DEPTH, PAT, wgs_per_cu, wrap ? "  L2-resident" : "", ms, (double)tiles * BM * ROWB / (ms * 1e-3) / 1e12);
}

int main() {
    const int rows = 36 * 22016, tiles = rows / BM;
    float *x, *out;
    CK(hipMalloc(&x, (size_t)rows * ROWB));
    CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(x, 0, (size_t)rows * ROWB));
    printf("%d rows of %d bytes (%.0f MB), %d tiles of %d rows\n", rows, ROWB, rows * (double)ROWB / 1e6, tiles, BM);
    for (int occ = 2; occ <= 4; ++occ) {
        run<128, 1, 0>(x, out, tiles, occ);
        run<128, 2, 0>(x, out, tiles, occ);
        run<256, 1, 0>(x, out, tiles, occ);
        run<256, 2, 0>(x, out, tiles, occ);
        run<512, 1, 0>(x, out, tiles, occ);
        run<1024, 1, 0>(x, out, tiles, occ);
        run<128, 1, 1>(x, out, tiles, occ);
        run<128, 2, 1>(x, out, tiles, occ);
        run<256, 1, 1>(x, out, tiles, occ);
        run<256, 2, 1>(x, out, tiles, occ);
        run<512, 1, 1>(x, out, tiles, occ);
        run<1024, 1, 1>(x, out, tiles, occ);
    }
    // the same instructions on data that stays in the L2 (32 tiles = 4 MB over the chip, 512 KB per XCD): the CU <- L2 rate
    for (int occ = 2; occ <= 4; ++occ) {
        run<128, 1, 0>(x, out, tiles, occ, 32);
        run<128, 2, 0>(x, out, tiles, occ, 32);
        run<256, 2, 0>(x, out, tiles, occ, 32);
        run<1024, 1, 1>(x, out, tiles, occ, 32);
        run<128, 1, 0>(x, out, tiles, occ, 256);
    }
    // which property of an instruction costs: the rows (cache lines) it touches, or how its 16-byte pieces lie within them?
    run<128, 1, 2>(x, out, tiles, 3, 32);
    run<256, 1, 2>(x, out, tiles, 3, 32);
    run<128, 1, 1>(x, out, tiles, 3, 32);     // 8 rows x 128 contiguous bytes per instruction
    run<256, 1, 1>(x, out, tiles, 3, 32);     // 4 rows x 256
    run<512, 1, 1>(x, out, tiles, 3, 32);     // 2 rows x 512
    run<128, 1, 2>(x, out, tiles, 3, 0);
    run<128, 1, 1>(x, out, tiles, 3, 0);
    return 0;
}
