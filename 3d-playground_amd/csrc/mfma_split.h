// fp32 products on the bf16 matrix cores: split operands.
//
// An fp32 number x is written as h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), each rounded to
// nearest.  The two subtractions are exact in fp32 (the subtrahend is the minuend rounded to fewer bits), and 3 x 8
// significand bits plus the sign of each residual cover fp32's 24: h + m + l == x exactly (tools/probes/split_probe.hip
// checks it on 2^20 values over 40 binades: maximum error 0).  A product of two such numbers is the sum of nine bf16 x bf16
// products, every one of them exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; the kernels issue the six
// largest,
//        a*b  ~=  ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm,
// and drop am*bl, al*bm, al*bl.  With |m| <= 2^-8 |x| and |l| <= 2^-16 |x| (round to nearest at each level) the dropped terms
// are at most 2^-23 |a*b| and 2^-25 |a*b| in the root mean square over random operands -- the size of the rounding an fp32
// accumulation step makes anyway (2^-24 |acc|), and the v_mfma_f32_32x32x16_bf16 adds its sixteen exact products before it
// rounds into the accumulator where the fp32 MFMA rounds after every two.  Measured against fp64 the kernels land within
// -15 .. +25 % of the v_mfma_f32_32x32x2_f32 kernels' error (DESIGN.md 4.6 has the tables for both modes).  Six bf16
// MFMAs of K = 16 take 6 x 32 cycles where eight fp32 MFMAs of K = 2 take 8 x 64: 2.7x the matrix rate, paid for with the
// vector ALU work below (4.5 instructions per element, the conversions and packed subtractions at half rate), which is what
// bounds the kernels that use it.
//
// Not IEEE in the corners: an infinite or NaN operand gives NaN (inf - inf in the residual), and residuals below the
// bf16 normal range are flushed by the matrix core.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Split8 { bf16x8 h, m, l; };

// Two values at a time: one v_cvt_pk_bf16_f32 for h and for m, the residuals by shift / mask + one packed subtraction.
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    const bf16x2 hp = {(__bf16)x0, (__bf16)x1};
    h = __builtin_bit_cast(unsigned, hp);
    // (the empty asm pins the PACKED value: without it the compiler derives h << 16 from a second, single-value conversion of
    // x0 -- seen in the weight-gradient kernel: 64 conversions per K-step instead of 32)
    asm("" : "+v"(h));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16);
    const float r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const bf16x2 mp = {(__bf16)r0, (__bf16)r1};
    m = __builtin_bit_cast(unsigned, mp);
    asm("" : "+v"(m));
    const float s0 = r0 - __builtin_bit_cast(float, m << 16);
    const float s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    // the last residual has at most 8 significant bits (x has 24, h and m took 8 each): its bf16 is its upper half, one
    // v_perm_b32 for the pair instead of a half-rate conversion
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}

__device__ __forceinline__ Split8 split8(const float (&x)[8]) {
    u32x4 hh, mm, ll;
#if (defined(RN_SPLIT_ABL) && RN_SPLIT_ABL == 1) || (defined(RN_KO) && (RN_KO & 8))                 // knock-out (timing only, wrong results): no vector ALU work
    Split8 k;
    k.h = __builtin_bit_cast(bf16x8, u32x4{__builtin_bit_cast(unsigned, x[0]), __builtin_bit_cast(unsigned, x[1]), __builtin_bit_cast(unsigned, x[2]), __builtin_bit_cast(unsigned, x[3])});
    k.m = __builtin_bit_cast(bf16x8, u32x4{__builtin_bit_cast(unsigned, x[4]), __builtin_bit_cast(unsigned, x[5]), __builtin_bit_cast(unsigned, x[6]), __builtin_bit_cast(unsigned, x[7])});
    k.l = k.h;
    return k;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b, c;
        split_pair(x[2 * j], x[2 * j + 1], a, b, c);
        hh[j] = a; mm[j] = b; ll[j] = c;
    }
    Split8 s;
    s.h = __builtin_bit_cast(bf16x8, hh);
    s.m = __builtin_bit_cast(bf16x8, mm);
    s.l = __builtin_bit_cast(bf16x8, ll);
    return s;
}

// Pre-split form of a packed fp32 tensor [rows][Kpad] (Kpad a multiple of 16): [rows][Kpad/16] records of 96 bytes, a
// record = the h, m and l terms (3 x 16 bf16) of the row's 16 values of that K-step -- what one K-step of a kernel reads of a
// row, contiguous.  Chunk i (8 consecutive values, i = index / 8) of the source goes to its record's three half-planes.
__device__ __forceinline__ void split_store_chunk(const float *__restrict__ src, void *__restrict__ dst, int64_t i) {
    const float4 p0 = reinterpret_cast<const float4 *>(src)[2 * i], p1 = reinterpret_cast<const float4 *>(src)[2 * i + 1];
    const float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const Split8 s = split8(v);
    char *rec = reinterpret_cast<char *>(dst) + (i >> 1) * 96 + (i & 1) * 16;
    *reinterpret_cast<bf16x8 *>(rec) = s.h;
    *reinterpret_cast<bf16x8 *>(rec + 32) = s.m;
    *reinterpret_cast<bf16x8 *>(rec + 64) = s.l;
}

// acc += a * b for one 32x32 tile and 16 values of k: the six products, smallest first.
#if (defined(RN_SPLIT_ABL) && RN_SPLIT_ABL == 2) || (defined(RN_KO) && (RN_KO & 4))                 // knock-out (timing only, wrong results): one MFMA of the six
#define RN_SPLIT_MFMA(ACC, A, B)                                                          \
    do {                                                                                  \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).l, (B).h, ACC, 0, 0, 0);        \
        ACC[0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, (A).h)[0] ^ __builtin_bit_cast(u32x4, (A).m)[1] ^ __builtin_bit_cast(u32x4, (B).m)[2] ^ __builtin_bit_cast(u32x4, (B).l)[3]) * 1e-30f; \
    } while (0)
#elif defined(RN_KO) && (RN_KO & 16)                  // shape experiment (timing only, wrong results): every 32x32x16 as two 16x16x32 (same pipe cycles)
typedef float rn_f32x4 __attribute__((ext_vector_type(4)));
#define RN_MF16(ACC, X, Y)                                                                \
    do {                                                                                  \
        rn_f32x4 lo_ = __builtin_shufflevector(ACC, ACC, 0, 1, 2, 3), hi_ = __builtin_shufflevector(ACC, ACC, 4, 5, 6, 7); \
        lo_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X, Y, lo_, 0, 0, 0);                \
        hi_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X, Y, hi_, 0, 0, 0);                \
        ACC[0] = lo_[0]; ACC[1] = lo_[1]; ACC[2] = lo_[2]; ACC[3] = lo_[3];               \
        ACC[4] = hi_[0]; ACC[5] = hi_[1]; ACC[6] = hi_[2]; ACC[7] = hi_[3];               \
    } while (0)
#define RN_SPLIT_MFMA(ACC, A, B)                                                          \
    do {                                                                                  \
        RN_MF16(ACC, (A).l, (B).h); RN_MF16(ACC, (A).h, (B).l); RN_MF16(ACC, (A).m, (B).m); \
        RN_MF16(ACC, (A).m, (B).h); RN_MF16(ACC, (A).h, (B).m); RN_MF16(ACC, (A).h, (B).h); \
    } while (0)
#else
#define RN_SPLIT_MFMA(ACC, A, B)                                                          \
    do {                                                                                  \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).l, (B).h, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).h, (B).l, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).m, (B).m, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).m, (B).h, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).h, (B).m, ACC, 0, 0, 0);        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16((A).h, (B).h, ACC, 0, 0, 0);        \
    } while (0)
#endif

// ------------------------------------------------------------------------------------------------------------------------------
// RN_FP32_SPLIT3 (round 5): TWO fp16 terms and THREE MFMAs per product instead of three bf16 terms and six.
//
// With a power-of-two scale s chosen so that the tensor's largest magnitude lands in [2^14, 2^15) (fp16: 11 significand bits, largest
// finite value 65504, smallest normal 2^-14), x*s = hi + lo with hi = f16(x*s), lo = f16(x*s - hi), both rounded to nearest: the
// subtraction is exact in fp32, |x*s - hi| <= 2^-11 |x*s|, so hi + lo carries 22 significand bits plus the residual's sign
// (|error| <= 2^-22 |x|, 2^-23.3 rms) for every element down to 2^-18 of the tensor's maximum; smaller elements keep an ABSOLUTE
// error of 2^-25 in scaled units = 2^-40 of the maximum (lo is then an fp16 subnormal, which the matrix core does not flush:
// tools/probes/f16_split_probe.hip).  A product is  ah*bh + ah*bl + al*bh  on v_mfma_f32_16x16x32_f16 / 32x32x16_f16 into the same
// fp32 accumulator (each term product exact, the sum of the instruction's 32 or 16 products rounded once); the dropped al*bl is at
// most 2^-22 of the product.  The result is the product of the SCALED operands: the epilogue multiplies by the two inverse scales
// (powers of two: exact).  Scales: per output channel for weights (rn_split_weights_f16 writes the inverse beside the terms), per
// tensor for activations and gradients from the amax word their producer's epilogue leaves (rn_conv_desc.x_amax / y_amax;
// rn_amax for tensors without one).  Against the three-term bf16 form: half the MFMAs, a two-plane instead of a three-plane weight
// image, 3 instead of 4.5 vector instructions per split value; error against fp64 measured per kernel in tools/fp32_mode_errors.py.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct SplitH8 { f16x8 h, l; };

// The scale of a tensor whose largest magnitude has the fp32 bit pattern `amax_bits` (sign cleared): 2^(14 - floor(log2 amax)), so that
// amax * scale is in [2^14, 2^15); 1 for an all-zero tensor.  Returned as the biased exponent field: scale = bits(se << 23), its inverse
// bits((254 - se) << 23); both stay normal numbers for every amax (tensors below 2^-112 are scaled by 2^126 and simply use less of fp16's
// range).  A NaN / infinite amax gives a finite scale: the non-finite elements themselves make the result NaN.
__host__ __device__ __forceinline__ int rn_f16_scale_exp(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);
    if (e == 0) return 127;
    const int se = 268 - e;                                  // 127 + 14 - (e - 127)
    return se > 253 ? 253 : se;
}
__host__ __device__ __forceinline__ int rn_f16_scale_exp_of(int e) {          // the same from the exponent field itself
    if (e <= 0) return 127;
    const int se = 268 - e;
    return se > 253 ? 253 : se;
}
__device__ __forceinline__ float rn_exp_to_float(int biased) { return __builtin_bit_cast(float, (unsigned)biased << 23); }

// Two values at a time: t = x*s; hi = cvt_pk_f16(t0, t1); lo = cvt_pk_f16(fma(x0, s, -hi0), fma(x1, s, -hi1)) -- the fma re-forms x*s
// exactly (a power-of-two scale) and subtracts hi in one rounding-free step.
__device__ __forceinline__ void split_pair_h(float x0, float x1, float s, unsigned &h, unsigned &l) {
    const f16x2 hp = {(_Float16)(x0 * s), (_Float16)(x1 * s)};
    h = __builtin_bit_cast(unsigned, hp);
    asm("" : "+v"(h));                                       // pin the PACKED conversion (see split_pair)
    const f16x2 hq = __builtin_bit_cast(f16x2, h);
    const float r0 = __builtin_fmaf(x0, s, -(float)hq[0]);
    const float r1 = __builtin_fmaf(x1, s, -(float)hq[1]);
    const f16x2 lp = {(_Float16)r0, (_Float16)r1};
    l = __builtin_bit_cast(unsigned, lp);
}

__device__ __forceinline__ SplitH8 split8h(const float (&x)[8], float s) {
    u32x4 hh, ll;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b;
        split_pair_h(x[2 * j], x[2 * j + 1], s, a, b);
        hh[j] = a; ll[j] = b;
    }
    SplitH8 r;
    r.h = __builtin_bit_cast(f16x8, hh);
    r.l = __builtin_bit_cast(f16x8, ll);
    return r;
}

// Pre-split fp16 form of a packed fp32 tensor [rows][Kpad] (Kpad a multiple of 16): [rows][Kpad/16] records of 64 bytes, a record =
// the hi and lo terms (2 x 16 fp16) of the row's 16 values of that K-step, scaled by the ROW's scale.
__device__ __forceinline__ void split_store_chunk_h(const float *__restrict__ src, void *__restrict__ dst, int64_t i, float s) {
    const float4 p0 = reinterpret_cast<const float4 *>(src)[2 * i], p1 = reinterpret_cast<const float4 *>(src)[2 * i + 1];
    const float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const SplitH8 sp = split8h(v, s);
    char *rec = reinterpret_cast<char *>(dst) + (i >> 1) * 64 + (i & 1) * 16;
    *reinterpret_cast<f16x8 *>(rec) = sp.h;
    *reinterpret_cast<f16x8 *>(rec + 32) = sp.l;
}

// acc += a * b for one 32x32 tile and 16 values of k: the three products, smallest first.
#define RN_SPLITH_MFMA(ACC, A, B)                                                         \
    do {                                                                                  \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16((A).l, (B).h, ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16((A).h, (B).l, ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16((A).h, (B).h, ACC, 0, 0, 0);         \
    } while (0)
// the same for one 16x16 tile and 32 values of k
#define RN_SPLITH_MFMA16(ACC, A, B)                                                       \
    do {                                                                                  \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16((A).l, (B).h, ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16((A).h, (B).l, ACC, 0, 0, 0);         \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16((A).h, (B).h, ACC, 0, 0, 0);         \
    } while (0)

// AMAX TABLES of a tensor (rn_conv_desc.x_amax / y_amax): per IMAGE, which binary exponents its elements have -- 256 bytes, byte e != 0
// iff some element's fp32 exponent field is e.  That is all a consumer needs (rn_f16_scale_exp takes the LARGEST exponent: the scale is
// a power of two), and it makes the producer side free of atomics and of read-backs:
//   * a producer lane that has stored values with largest magnitude v writes the byte  table[image][exponent(v)] = 1  -- a plain store,
//     idempotent, no ordering between lanes, waves or launches needed (the parity classes of a stride-2 data gradient, an accumulation
//     in place: later launches just add bytes); lines written on several XCDs merge byte-wise when the kernel ends;
//   * a consumer wave reads the image's 256 bytes as one dword per lane and takes the highest non-zero byte (rn_amax_exp).
// Per image, not per tensor, so that the scales -- and with them every bit of an image's result -- do not depend on which other images
// share the launch (tests/test_gpu_batch8.py pins that); exact and deterministic: the largest exponent present, nothing else.
// History of the round (profiles/r05_amax_protocol_ab.txt): one word per image raised by atomic max -- filtered by a read-back, at
// device scope or inside the XCD's L2, per wave or per workgroup -- cost 7 to 20 % of the training step: the workgroups of a launch's
// first round all read the word before any has raised it and their same-address atomics retire one after the other, and a read-back
// early in an epilogue holds up every later load of the wave (vmcnt returns in order); fire-and-forget atomics were 2x slower still.
#define RN_AMAX_BYTES 256
__device__ __forceinline__ float rn_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// producer: this lane stored values of largest magnitude v in image n
#ifndef RN_AMAX_KO               // knock-out (timing only, wrong scales downstream): producers write nothing
#define RN_AMAX_KO 0
#endif
__device__ __forceinline__ void rn_amax_note(void *tables, int64_t n, float v) {
    if (tables == nullptr || RN_AMAX_KO) return;
    reinterpret_cast<unsigned char *>(tables)[n * RN_AMAX_BYTES + ((__builtin_bit_cast(unsigned, v) >> 23) & 0xffu)] = 1;
}
// consumer: the largest exponent field present in image n's table; all 64 lanes of the wave take part, the result is wave-uniform
__device__ __forceinline__ int rn_amax_exp(const void *tables, int64_t n) {
    const unsigned d = reinterpret_cast<const unsigned *>(reinterpret_cast<const unsigned char *>(tables) + n * RN_AMAX_BYTES)[threadIdx.x & 63];
    int e = d ? 4 * (int)(threadIdx.x & 63) + ((31 - __builtin_clz(d)) >> 3) : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(e, off, 64); e = o > e ? o : e; }
    return __builtin_amdgcn_readfirstlane(e);
}
// the same over images 0 .. count-1 (the weight gradient: one scale for a reduction that runs over all images)
__device__ __forceinline__ int rn_amax_exp_all(const void *tables, int count) {
    int e = 0;
    for (int i = 0; i < count; ++i) {
        const unsigned d = reinterpret_cast<const unsigned *>(reinterpret_cast<const unsigned char *>(tables) + (int64_t)i * RN_AMAX_BYTES)[threadIdx.x & 63];
        const int ei = d ? 4 * (int)(threadIdx.x & 63) + ((31 - __builtin_clz(d)) >> 3) : 0;
        e = ei > e ? ei : e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(e, off, 64); e = o > e ? o : e; }
    return __builtin_amdgcn_readfirstlane(e);
}
// every lane gets the exponent of ITS OWN table (table_lane = tables + image * RN_AMAX_BYTES; nullptr: 0) -- the lanes of a wave lie in
// one or two images, rarely more; all 64 lanes take part
__device__ __forceinline__ int rn_amax_exp_lanes(const void *table_lane) {
    const unsigned long long me = (unsigned long long)(uintptr_t)table_lane;
    int mine = 0;
    unsigned long long pending = __ballot(table_lane != nullptr);
    while (pending != 0ull) {
        const int first = __builtin_ctzll(pending);
        const unsigned long long p0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(me >> 32), first) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)me, first);
        const int e0 = rn_amax_exp(reinterpret_cast<const void *>((uintptr_t)p0), 0);
        if (me == p0) mine = e0;
        pending &= ~__ballot(me == p0);
    }
    return mine;
}
// a word (fp32 bit pattern) with exponent field e and a full mantissa: what the derived (Winograd-domain) words are made of
__host__ __device__ __forceinline__ unsigned rn_amax_word_of_exp(int e, int gain_log2) {
    if (e <= 0) return 0u;
    const int g = e + gain_log2 > 254 ? 254 : e + gain_log2;
    return ((unsigned)g << 23) | 0x7fffffu;
}

// One weight row of the fp16 pre-split form, by one wave: the row's largest magnitude (one pass, wave maximum), its power-of-two scale,
// then the hi / lo terms of every 8-value chunk (rn_split_weights_f16; rn_prep_batched job kind 5).
__device__ __forceinline__ void split_row_f16(const float *__restrict__ src, void *__restrict__ dst, float *__restrict__ unscale,
                                              int64_t row, int Kpad, int lane) {
    const float4 *r4 = reinterpret_cast<const float4 *>(src + row * Kpad);
    float am = 0.f;
    for (int i = lane; i < Kpad / 4; i += 64) {
        const float4 q = r4[i];
        am = fmaxf(fmaxf(am, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
    }
    const int se = rn_f16_scale_exp(__builtin_bit_cast(unsigned, rn_wave_max(am)));
    if (lane == 0) unscale[row] = rn_exp_to_float(254 - se);
    const float sc = rn_exp_to_float(se);
    const int64_t c0 = row * (Kpad / 8);
    for (int i = lane; i < Kpad / 8; i += 64) split_store_chunk_h(src, dst, c0 + i, sc);
}
