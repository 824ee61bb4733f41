// Probe (round 4): what does a 256 x 256 x 64 bf16 tile with EIGHT waves (2 x 4, each 128 x 64 on v_mfma_f32_16x16x32_bf16), operands by
// direct-to-LDS DMA one K-tile ahead, fragment reads one phase ahead and ONE barrier per K-tile reach on the dominant layer's GEMM shape
// (M = 8 x 135 x 240 = 259 200, N = 256, K = 2304)?  The bf16 engine's kernels (conv_bf16.hip: 128 x 128 ... 256 x 256 tiles, a barrier per
// 64-byte K-step) all sit at 750-810 TFLOP/s there whatever the tile (profiles/r04_bf16_tile_variants.txt); the programming guide's
// eight-phase template reports 1 320-1 470 on random operands.  Plain GEMM C[M][N] = A[M][K] . B[N][K]^T, rows K-contiguous (= NHWC
// activations of a 1x1 layer and packed weights).  Build + run:
//   hipcc -O3 --offload-arch=gfx950 -o gemm8_probe gemm8_probe.hip && ./gemm8_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

__device__ __forceinline__ v4i32 make_rsrc(const void *base, unsigned bytes) {
    const uint64_t b = (uint64_t)base;
    v4i32 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
// lane l's 16 bytes at base + voff + soff -> LDS lds_dst + 16 l (out of range: zeros).  asm: the compiler's waitcnt pass must not see it.
__device__ __forceinline__ void dma16(v4i32 rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// LDS image of an operand tile: [256 rows][128 bytes = 64 bf16 of k], 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7).  A 16x16x32
// fragment read takes row lane & 15 of a 16-row block, chunk (lane >> 4) + 4 kh: the four 16-lane groups ds_read_b128 is served in
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32) then hit 16 different 16-byte slots of the 256-byte bank row (checked by enumeration).
constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128;
constexpr int OPB = 256 * ROWB;                 // bytes of one operand tile: 32 KB
constexpr int BUFB = 2 * OPB;                   // one K-tile buffer: A then B
__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const __bf16 *__restrict__ A, const __bf16 *__restrict__ B, __bf16 *__restrict__ C,
                                                     int M, int N, int K, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wr = wave >> 2, wc = wave & 3;    // 2 x 4 waves: rows wr * 128, columns wc * 64
    const int ntn = N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int nkt = K / BK;
    const int64_t a_bytes = VARIANT >= 1 ? (int64_t)(M + 482) * ldc * 2 : (int64_t)M * K * 2;
    const v4i32 rs_a = make_rsrc(A, (unsigned)(a_bytes > 0x7FFFFFFF ? 0x7FFFFFFF : a_bytes));
    const v4i32 rs_b = make_rsrc(B, (unsigned)((int64_t)N * K * 2));
    const unsigned lds0 = lds_addr(lds);

    // ---- staging: instruction q (0..63) of a K-tile fills 8 rows x 128 bytes: q < 32 -> A rows 8 q .., else B rows 8 (q - 32) ..; wave w
    // issues q = 8 i + w' ... : its 8 instructions i = 0..7 are q = w + 8 i (4 of A, 4 of B).  Lane -> row (lane >> 3), slot (lane & 7):
    // it fetches chunk slot ^ swz(row).
    // row of instruction q = wave + 8 i: (q & 31) * 8 + (lane >> 3) = 64 i + 8 wave + (lane >> 3) (i < 4: A, else B, i - 4): the swizzle of the
    // row does not depend on i, so ONE lane offset per operand and the i-th instruction adds 64 rows on the scalar offset.  Rows past the end
    // of an operand are past its descriptor's range (zero-fill).
    const int row0 = 8 * wave + (lane >> 3);
    const int chunk = (lane & 7) ^ swz(row0);
    const int lda = VARIANT >= 1 ? ldc : K;
    const unsigned voff_a = (unsigned)(((int64_t)(m0 + row0) * lda) * 2 + chunk * 16);
    const unsigned voff_b = (unsigned)(((int64_t)(n0 + row0) * K) * 2 + chunk * 16);
    auto dma = [&](const int i, const int kt, const int buf) {        // instruction i of this wave for K-tile kt into buffer buf
        const int q = wave_u + 8 * i;
        int ktr = kt;
        if (VARIANT == 2) ktr = (kt & ~3) | ((kt + tile) & 3);       // the four 64-channel chunks of a tap in an order rotated by the tile
        unsigned so;
        if (i >= 4) so = (unsigned)((ktr * BK + (i - 4) * 64 * K) * 2);
        else if (VARIANT == 0) so = (unsigned)((ktr * BK + i * 64 * K) * 2);
        else {
            // a 3x3 convolution's activation operand: K = 9 taps x 256 channels of a [rows][ldc] tensor, tap (r, s) reads pixel row
            // m + (r - 1) * 240 + (s - 1) -- nine overlapping passes over the same 133 MB instead of one pass over 1.19 GB
            const int tap = ktr >> 2, c0 = (ktr & 3) * 64;
            so = (unsigned)((((tap / 3 - 1) * 240 + (tap % 3 - 1) + 241 + i * 64) * ldc + c0) * 2);
        }
        so = __builtin_amdgcn_readfirstlane(so);
        const v4i32 rsel = i < 4 ? rs_a : rs_b;
        v4i32 r_;
        r_.x = __builtin_amdgcn_readfirstlane(rsel.x); r_.y = __builtin_amdgcn_readfirstlane(rsel.y);
        r_.z = __builtin_amdgcn_readfirstlane(rsel.z); r_.w = __builtin_amdgcn_readfirstlane(rsel.w);
        dma16(r_, lds0 + (unsigned)(buf * BUFB + q * 1024), i < 4 ? voff_a : voff_b, so);
    };

    // ---- fragments: A block rb (16 rows) of this wave's 128 rows, k half kh; B block cb of its 64 columns
    const int lr = lane & 15, lg = lane >> 4;
    // one address per operand and lane: the block index adds 16 rows = 2048 bytes (16 rows do not change (r >> 1) & 7), the k half flips
    // chunk bit 2 = 64 bytes
    const int ra = wr * 128 + lr, rbb = wc * 64 + lr;
    const int a0 = ra * ROWB + 16 * (lg ^ swz(ra));
    const int b0 = OPB + rbb * ROWB + 16 * (lg ^ swz(rbb));
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa0[4][2], fa1[4][2], fb0[2][2], fb1[2][2];           // A quadrant 0 / 1 (64 rows each), B quadrant 0 / 1 (32 columns each)
    auto rd = [&](const char *S, int off) { return *reinterpret_cast<const bf16x8 *>(S + off); };
    auto read_a = [&](bf16x8 (&f)[4][2], const char *S, int qm) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) f[i][kh] = rd(S, (a0 ^ (64 * kh)) + (4 * qm + i) * 2048);
    };
    auto read_b = [&](bf16x8 (&f)[2][2], const char *S, int qn) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) f[j][kh] = rd(S, (b0 ^ (64 * kh)) + (2 * qn + j) * 2048);
    };
    // weight fragment first: a lane then holds 4 consecutive columns (N) of one row -- 8-byte bf16 stores in the epilogue
    auto mma = [&](const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2], int qm, int qn) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * qm + i][2 * qn + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kh], fa[i][kh], acc[4 * qm + i][2 * qn + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: K-tile 0 -> buffer 0 (all 8), K-tile 1 -> buffer 1 (all 8); wait for tile 0, read its first fragments
#pragma unroll
    for (int i = 0; i < 8; ++i) dma(i, 0, 0);
    if (nkt > 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma(i, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    read_a(fa0, lds, 0);
    read_b(fb0, lds, 0);

    // ---- main loop: K-tile t in buffer t & 1.  Before the barrier of tile t: its buffer is no longer read by this wave (all fragments of
    // the tile are in registers after phase 1's issue ... see the order below) and tile t + 1 has landed.  After it: buffer t & 1 is free
    // for tile t + 2 (3 + 3 + 2 instructions over the next three phases), buffer (t + 1) & 1 is readable.
    // The B fragment sets swap roles every K-tile (x = the set whose fragments are already there, y = the other): the order of the four
    // quadrants is (0,x) (0,y) (1,y) (1,x), and phase 3 -- which still needs x and A set 1 -- loads the NEXT tile's A set 0 and its B
    // quadrant qy into the y set: the next tile starts from (0, y).  No staging registers.
    auto body = [&](const int t, bf16x8 (&fbx)[2][2], bf16x8 (&fby)[2][2], const int qx, const int qy) {
        const char *S = lds + (t & 1) * BUFB;
        const char *Sn = lds + ((t + 1) & 1) * BUFB;
        const bool more = t + 1 < nkt, pre = t > 0 && more;
        read_b(fby, S, qy);
        mma(fa0, fbx, 0, qx);
        if (pre) { dma(3, t + 1, (t + 1) & 1); dma(4, t + 1, (t + 1) & 1); dma(5, t + 1, (t + 1) & 1); }
        read_a(fa1, S, 1);
        mma(fa0, fby, 0, qy);
        if (pre) { dma(6, t + 1, (t + 1) & 1); dma(7, t + 1, (t + 1) & 1); }
        mma(fa1, fby, 1, qy);
        // every fragment of tile t is in registers (the compiler's lgkmcnt waits sit in front of the MFMAs that use them); tile t + 1 has
        // been requested completely at least one phase ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (more) {
            read_a(fa0, Sn, 0);
            read_b(fby, Sn, qy);
        }
        mma(fa1, fbx, 1, qx);
        if (t + 2 < nkt) { dma(0, t + 2, t & 1); dma(1, t + 2, t & 1); dma(2, t + 2, t & 1); }
    };
    for (int t = 0; t < nkt; t += 2) {
        body(t, fb0, fb1, 0, 1);
        if (t + 1 < nkt) body(t + 1, fb1, fb0, 1, 0);
    }

    // ---- epilogue (probe): straight from the accumulators, 4 consecutive columns of one row per lane and block
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        const int64_t m = (int64_t)m0 + wr * 128 + rb * 16 + lr;
        if (m < M) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int n = n0 + wc * 64 + cb * 16 + 4 * lg;
                if (n < N) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[rb][cb][e];
                    *reinterpret_cast<bf16x4 *>(C + m * N + n) = o;
                }
            }
        }
    }
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 259200, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 2304;
    printf("gemm8 probe: M %d N %d K %d (bf16 in, fp32 accumulate, bf16 out)\n", M, N, K);
    std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) & 0xFFFF) / 32768.0f - 1.0f; };   // uniform [-1, 1)
    for (auto &v : ha) v = f2bf(rnd());
    for (auto &v : hb) v = f2bf(rnd() * 0.05f);
    __bf16 *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, ha.size() * 2)); CHECK(hipMalloc(&dB, hb.size() * 2)); CHECK(hipMalloc(&dC, (size_t)M * N * 2));
    CHECK(hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(dC, 0, (size_t)M * N * 2));
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    const size_t ldsb = 2 * BUFB;
    CHECK(hipFuncSetAttribute((const void *)gemm8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const int variant = argc > 4 ? atoi(argv[4]) : 0;
    const int ldc = argc > 5 ? atoi(argv[5]) : 256;      // variants 1, 2: elements between pixel rows of the activation tensor
    CHECK(hipFuncSetAttribute((const void *)gemm8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    CHECK(hipFuncSetAttribute((const void *)gemm8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    auto launch = [&]() {
        if (variant == 1) hipLaunchKernelGGL(gemm8_kernel<1>, dim3(tiles), dim3(512), ldsb, 0, dA, dB, dC, M, N, K, ldc);
        else if (variant == 2) hipLaunchKernelGGL(gemm8_kernel<2>, dim3(tiles), dim3(512), ldsb, 0, dA, dB, dC, M, N, K, ldc);
        else hipLaunchKernelGGL(gemm8_kernel<0>, dim3(tiles), dim3(512), ldsb, 0, dA, dB, dC, M, N, K, ldc);
    };
    if (variant >= 1) printf("variant 1: conv-like activation addressing (K = 9 x 256 over a [M + 482][256] tensor); the numeric check is skipped\n");
    launch();
    CHECK(hipDeviceSynchronize());
    // check a sample of outputs against fp64 on the host
    std::vector<uint16_t> hc((size_t)M * N);
    CHECK(hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    int bad = 0;
    for (int it = 0; it < (variant >= 1 ? 0 : 4000); ++it) {
        const int m = (int)(((uint64_t)it * 2654435761u) % (uint64_t)M), n = (it * 97) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(ha[(size_t)m * K + k]) * bf2f(hb[(size_t)n * K + k]);
        const double got = bf2f(hc[(size_t)m * N + n]);
        const double err = fabs(got - ref) / (fabs(ref) + 0.05);
        if (err > worst) worst = err;
        if (err > 2e-2) { if (bad < 5) printf("  mismatch at (%d, %d): got %g want %g\n", m, n, got, ref); ++bad; }
    }
    // the last row tile and a full tile, every element of a few rows
    for (int m : {0, 1, 255, 256, M - 1, M - 129}) {
        if (variant >= 1) break;
        for (int n = 0; n < N; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf2f(ha[(size_t)m * K + k]) * bf2f(hb[(size_t)n * K + k]);
            const double err = fabs(bf2f(hc[(size_t)m * N + n]) - ref) / (fabs(ref) + 0.05);
            if (err > worst) worst = err;
            if (err > 2e-2) ++bad;
        }
    }
    printf("check: worst relative error %.3e, %d mismatches\n", worst, bad);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 5; ++i) launch();
        CHECK(hipEventRecord(e0));
        const int iters = 30;
        for (int i = 0; i < iters; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters;
        printf("  %.4f ms  %.1f TFLOP/s (%.1f %% of 2.5 PF)\n", ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 25.0);
    }
    return bad ? 1 : 0;
}
