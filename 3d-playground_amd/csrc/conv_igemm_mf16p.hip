// Short-reduction split-operand GEMM as ONE CONTINUOUS K-STEP STREAM per workgroup: persistent workgroups, two accumulator sets,
// the stores of tile n spread over the K-steps of tile n + 1 (round 4).  The plain-GEMM form of the Winograd stage (36 positions x T
// tiles, K = Cin = 128 .. 512, conv.wino_conv_group) -- 13 ms of the 80 ms training step -- is its customer.
//
// What round 3 measured on these launches (profiles/r03_split_clock.txt 6): t = 0.082 ms + 0.61 us x K at M x N = 259 200 x 256; the
// fixed part is the output's stores plus a tile's set-up, and it ADDS to the K loops instead of hiding under the other two workgroups
// of the CU (knock-out: no stores 0.201 of 0.275 ms).  Round 4, first attempt (tools/probes/mf16_direct_epilogue_persist_v1.hip.txt,
// profiles/r04_persist_ab_micro.txt): persistent workgroups that request tile n + 1's first operands before tile n's stores, stores
// straight from the accumulators -- correct, and no faster (0.745 against 0.712 ms on the Winograd GEMM, the training step 96.5 against
// 99.5 images/s): the burst of a tile's 64 KB of stores is still a burst.  A CU's vector-memory pipe is one in-order queue: while a
// workgroup's stores drain at the CU's share of the HBM write rate, the operand loads of the other workgroups of the CU sit behind them
// and their K loops wait; and because every CU alternates the same two phases the chip falls into step -- all storing, then all
// computing.  So the stores have to stop being a burst:
//   * a workgroup keeps TWO accumulator sets (2 x 64 registers; ~200 registers, two workgroups per CU) and walks its tiles as one
//     stream of K-steps: while tile n + 1 accumulates into one set, the other set -- tile n -- is stored in slices, 16 / nks store
//     instructions per K-step right behind that step's operand loads.  A CU then writes at a steady ~16 KB per K-step, far under its
//     share of the memory system, at every moment of the launch;
//   * the stream has no tile boundary: the "next step" whose operands a K-step requests is step 0 of the NEXT tile when the current
//     tile ends (its addresses are computed at the top of the tile's last step), so the matrix pipe sees an unbroken sequence;
//   * stores go straight from the registers: the MFMAs run with the weight fragment first and the activation fragment second, which
//     leaves in a lane four consecutive output channels of one pixel = a 16-byte store (16 rows x 64 bytes per instruction, the
//     neighbouring column block completes the 128-byte lines); buffer stores, out-of-range offset for rows past M;
//   * tiles are dealt statically: XCD x owns a contiguous range, workgroup j of its gridDim / 8 takes tiles j, j + gridDim / 8, ...
// Conditions (the launcher's): raw epilogue (plain GEMM: no scale / shift / addend / mask / activation), pre-split weights, Cin a
// multiple of 64 (an even number of 32-deep K-steps: every tile starts on LDS buffer 0), Cout a multiple of 4 and > 64, 1x1, dense output
// below 4 GiB.  Same arithmetic as conv_igemm_mf16.hip (same six products in the same order into the same accumulator chain).
#include "conv_igemm_tile.h"
#include "conv_wgrad_geom.h"

typedef float f32x4a __attribute__((ext_vector_type(4)));

#define RN_SPLIT_MFMA16W(ACC, A, W)                                                       \
    do {                                                                                  \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W).h, (A).l, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W).l, (A).h, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W).m, (A).m, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W).h, (A).m, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W).m, (A).h, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W).h, (A).h, ACC, 0, 0, 0);        \
    } while (0)

__global__ __launch_bounds__(256, 2) void conv_igemm_mf16_pipe_kernel(const rn_conv_desc d, const int ntiles, const float *__restrict__ x,
                                                                   const float *__restrict__ w, float *__restrict__ y) {
    constexpr int BK = 32, BM = 128, BN = 128, NSN = BN / 16;
    constexpr int BPL = BN * 16;                           // floats' worth of one bf16 plane: rows x 64 bytes
    constexpr int BSTEP = 3 * BPL;                         // one buffer: B planes h, m, l
    constexpr int IB = 3 * BN / 16 / 4;                    // direct-to-LDS instructions per step and wave
    constexpr int NST = 2 * NSN;                           // 16-byte store instructions per wave and tile
    __shared__ float lds[2 * BSTEP];

    // ---- this workgroup's tiles (see the header): start, stride, end
    int tile, tend;
    const int tstride = gridDim.x >> 3;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int cnt = q + (xcd < r ? 1 : 0);
        if (j >= cnt) return;                              // (whole workgroup: more slots than this XCD has tiles)
        tile = start + j;
        tend = start + cnt;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int ntn = (d.Cout + BN - 1) / BN;
    const int HoWo = d.Ho * d.Wo;
    const int64_t M = (int64_t)d.N * HoWo;
    const int K = d.Cin;                                   // 1x1
    const int nks = K / BK;                                // even (launcher)
    const unsigned lds0 = lds_addr(lds);
    const int fb0 = Mf16Geom::read_addr(lane, 0) / 4;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(y, (short)0, (int)(unsigned)(M * d.Cout * 4), 0x00020000);
    const int q_st = nks >= NST ? 1 : NST / nks;           // stores per K-step (nks = 2, 4, 8, 16 -> 8, 4, 2, 1)

    // ---- operand addresses of the tile whose steps are being REQUESTED (one step ahead of the tile being multiplied)
    int m0 = 0, n0 = 0, img = 0;                           // img: the "image" (= Winograd position) the tile's rows start in
    int a_base[2];
    unsigned b_voff[IB];
    auto setup = [&](const int t) {
        m0 = (t / ntn) * BM;
        n0 = (t % ntn) * BN;
        const int n_first = __builtin_amdgcn_readfirstlane((int)(m0 / HoWo));   // the "image" = Winograd position: its own weight matrix
        img = n_first;                                     // (the descriptors are built from it where they are used: carried through the
                                                           // K-step loop as descriptors, the compiler parks them in vector registers)
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {                   // this lane's two activation rows: 1x1, stride a, no padding taps
            const int row = 32 * wave + 16 * sm + lr;
            unsigned off = 0x80000000u;
            if ((int64_t)m0 + row < M) {
                const unsigned rel = (unsigned)(m0 - n_first * HoWo + row);
                const unsigned n = rel / (unsigned)HoWo;
                const unsigned rem = rel - n * (unsigned)HoWo;
                const unsigned oh = rem / (unsigned)d.Wo, ow = rem - oh * (unsigned)d.Wo;
                const int ih = (int)oh * d.a + d.p, iw = (int)ow * d.a + d.p_w;
                if (((ih | iw) >= 0) & (ih < d.Hi) & (iw < d.Wi))
                    off = (unsigned)((int)((int64_t)n * d.x_batch_stride * 4) + ((ih * d.Wi + iw) * d.Cin + 8 * lg) * 4);
            }
            a_base[sm] = (int)off;
        }
#pragma unroll
        for (int j = 0; j < IB; ++j) {
            const int q = wave * IB + j, plane = q / (BN / 16), brow = (q % (BN / 16)) * 16 + (lane >> 2);
            const int c = Mf16Geom::dma_chunk(lane);
            const int n = n0 + brow;
            b_voff[j] = n < d.Cout ? (unsigned)(n * K * 6 + (c >> 1) * 96 + plane * 32 + (c & 1) * 16) : 0x80000000u;
        }
    };
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    struct ARegs { f32x4v v[4]; };
    auto load_a = [&](ARegs &ar, const int ks) {           // an out-of-range base stays out of range with the step's offset in soffset
        const int64_t x_floats = ((int64_t)d.N - 1 - img) * d.x_batch_stride + (int64_t)d.Hi * d.Wi * d.Cin;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (int64_t)img * d.x_batch_stride), (short)0,
            (int)(unsigned)(x_floats * 4 > 0x7FFFFFFF ? 0x7FFFFFFF : x_floats * 4), 0x00020000);
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            ar.v[2 * sm] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_base[sm], ks * BK * 4, 0));
            ar.v[2 * sm + 1] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)((unsigned)a_base[sm] + 16u), ks * BK * 4, 0));
        }
    };
    auto dma_b = [&](const int ks, const int buf) {
        unsigned so = (unsigned)(ks * 192);
        asm volatile("" : "+s"(so));                       // a register, never a literal (soffset takes an SGPR or an inline constant)
        const v4i32 rb_ = make_rsrc(reinterpret_cast<const char *>(w) + (int64_t)img * d.w_batch_stride * 6, (unsigned)((int64_t)d.Cout * K * 6));
#pragma unroll
        for (int j = 0; j < IB; ++j)
            dma16(rb_, lds0 + (unsigned)(buf * BSTEP * 4 + (wave_u * IB + j) * 1024), b_voff[j], so);
    };

    // ---- stores of a FINISHED tile (its m0 / n0 in pm0 / pn0): instruction i of NST = accumulator [i / NSN][i % NSN], i.e. output row
    // pm0 + 32 wave + 16 (i / NSN) + lr, channels pn0 + 16 (i % NSN) + 4 lg .. + 3.  i0 <= i < i1 (workgroup-uniform scalars): the
    // slice of this K-step.  Every index is a compile-time register; the bounds test is a scalar branch.
    int pm0 = 0, pn0 = 0;
    auto store_slice = [&](const f32x4a (&acc)[2][NSN], const int i0, const int i1) {
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
            const int64_t mr = (int64_t)pm0 + 32 * wave + 16 * sm + lr;
            const unsigned row_off = mr < M ? (unsigned)((mr * d.Cout + pn0 + 4 * lg) * 4) : 0xFFFFFFFFu;
#pragma unroll
            for (int sn = 0; sn < NSN; ++sn) {
                const int i = sm * NSN + sn;
                if (i >= i0 && i < i1) {
                    const bool ok = (row_off != 0xFFFFFFFFu) && (pn0 + 16 * sn + 4 * lg < d.Cout);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[sm][sn]), rs_y, (int)(ok ? row_off + 64u * sn : 0xFFFFFFFFu), 0, 0);
                }
            }
        }
    };

    // ---- one tile: nks K-steps into `cur`, the slices of `prev` (the tile before, if any) stored along the way.  `areg` holds the A
    // values of the step about to be multiplied; each step requests the next one -- the next TILE's step 0 from the last step.
    ARegs areg;
    bool more = true;                                      // is there a tile after the one being multiplied
    auto run_tile = [&](f32x4a (&cur)[2][NSN], const f32x4a (&prev)[2][NSN], const bool have_prev) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NSN; ++j) cur[i][j] = f32x4a{0.f, 0.f, 0.f, 0.f};
        const int my_m0 = m0, my_n0 = n0;                  // the tile being multiplied (setup() moves m0 / n0 on in its last step)
        auto k_step = [&](const int ks, const int rb, const bool last) {
            asm volatile("" : "+v"(areg.v[0]), "+v"(areg.v[1]), "+v"(areg.v[2]), "+v"(areg.v[3]));
            Split8 sa[2];
#pragma unroll
            for (int sm = 0; sm < 2; ++sm) {
                const float av[8] = {areg.v[2 * sm][0], areg.v[2 * sm][1], areg.v[2 * sm][2], areg.v[2 * sm][3],
                                     areg.v[2 * sm + 1][0], areg.v[2 * sm + 1][1], areg.v[2 * sm + 1][2], areg.v[2 * sm + 1][3]};
                sa[sm] = split8(av);
            }
            int nxt = ks + 1;
            if (last) {                                    // workgroup-uniform: the stream moves on to the next tile
                tile += tstride;
                more = tile < tend;
                if (more) setup(tile);
                nxt = 0;                                   // (no next tile: step 0 of this one again -- loaded, never used)
            }
            dma_b(nxt, rb ^ 1);
            load_a(areg, nxt);
            if (have_prev) store_slice(prev, ks * q_st, ks * q_st + q_st);
            const float *S = lds + rb * BSTEP + fb0;
#pragma unroll
            for (int sn = 0; sn < NSN; ++sn) {
                Split8 sb;
                sb.h = *reinterpret_cast<const bf16x8 *>(S + 256 * sn);
                sb.m = *reinterpret_cast<const bf16x8 *>(S + 256 * sn + BPL);
                sb.l = *reinterpret_cast<const bf16x8 *>(S + 256 * sn + 2 * BPL);
#pragma unroll
                for (int sm = 0; sm < 2; ++sm) RN_SPLIT_MFMA16W(cur[sm][sn], sa[sm], sb);
            }
            RN_PIN();
            rn_wait_dma();                                 // next step's planes landed and A registers arrived; this step's stores done
            __syncthreads();
        };
        for (int ks = 0; ks < nks; ks += 2) {
            k_step(ks, 0, false);
            k_step(ks + 1, 1, ks + 2 == nks);
        }
        if (have_prev) store_slice(prev, nks * q_st, NST);  // what the slices did not cover (nks not a power of two; none otherwise)
        pm0 = my_m0;
        pn0 = my_n0;
    };

    f32x4a accA[2][NSN], accB[2][NSN];
    setup(tile);
    load_a(areg, 0);
    dma_b(0, 0);
    rn_wait_dma();
    __syncthreads();
    bool have_prev = false;
    for (;;) {
        run_tile(accA, accB, have_prev);
        if (!more) { store_slice(accA, 0, NST); break; }
        run_tile(accB, accA, true);
        have_prev = true;
        if (!more) { store_slice(accB, 0, NST); break; }
    }
}

// resident slots: two workgroups per CU (two accumulator sets: ~200 registers), a multiple of 8; RN_OPT_PERSIST_WGS overrides (the parity
// tests force a few workgroups onto small problems so that each walks several tiles)
static int pipe_slots() {
    const int forced = rn_get_option(RN_OPT_PERSIST_WGS);
    if (forced > 0) return (forced + 7) / 8 * 8;
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        slots = 2 * cus / 8 * 8;
    }
    return slots;
}

// -> true if launched (the plain-GEMM form only: variant 0 of rn_igemm_split_launch).
bool rn_igemm_mf16_pipe_launch(int variant, const rn_conv_desc *d, const float *x, const float *w, float *y, hipStream_t s, int *rc) {
    if (variant != 0 || !rn_get_option(RN_OPT_PERSIST) || !rn_get_option(RN_OPT_MF16)) return false;
    if (d->w_format != 1 || d->kh != 1 || d->kw != 1 || (d->Cin % 64) != 0 || d->div_shift != 0 || d->in_relu) return false;
    if (d->Cin > rn_get_option(RN_OPT_PERSIST_MAX_K) || (d->Cout % 4) != 0 || d->Cout <= 64) return false;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
    if (M * d->Cout * 4 >= 0xFFFFFFF0LL) return false;                       // the output as one buffer with 32-bit offsets
    const int64_t tiles = ((M + 127) / 128) * ((d->Cout + 127) / 128);
    const int slots = pipe_slots();
    if (tiles <= slots || tiles > 0x7fffffff) return false;                 // fewer tiles than slots: nothing to pipeline
    hipLaunchKernelGGL(conv_igemm_mf16_pipe_kernel, dim3((unsigned)slots), dim3(256), 0, s, *d, (int)tiles, x, w, y);
    const hipError_t e = hipGetLastError();
    *rc = e == hipSuccess ? RN_OK : (int)e;
    return true;
}
