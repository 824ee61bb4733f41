// Host-side range check of the weight-gradient kernels' LDS indexing (see 3d-playground_amd/csrc/conv_wgrad_geom.h).
// Enumerates every wave, lane, K-step and instruction of every tile instance the launchers use; exits non-zero and prints
// the first violation.  Built and run by tests/test_wgrad_index_ranges.py (g++, no GPU).
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "conv_wgrad_geom.h"

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (fails++ < 10) { std::printf("FAIL " __VA_ARGS__); std::printf("\n"); } } } while (0)

template <int WM, int WN, int WK>
static void check_fp32(const char *name) {
    using G = WgradGeom<WM, WN, WK>;
    // pixel table: every (K-step, wave, lane, instruction) inside one 256-entry half, and per K-step the 4 waves x lanes x
    // instructions read exactly the step's WK entries
    for (int ks = 0; ks < 4 * G::TB; ++ks) {
        std::set<int> seen;
        for (int w = 0; w < 4; ++w)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < G::IB; ++j) {
                    const int i = G::tab_index(ks, w, l, j);
                    CHECK(i >= 0 && i < G::TAB, "%s: table index %d (ks %d wave %d lane %d j %d) outside [0,%d)", name, i, ks, w, l, j, G::TAB);
                    CHECK(i / WK == ks % G::TB, "%s: table index %d belongs to another K-step than %d", name, i, ks);
                    seen.insert(i);
                }
        CHECK((int)seen.size() == WK, "%s: K-step %d touches %d table entries, expected %d", name, ks, (int)seen.size(), WK);
    }
    // DMA destinations: the 1 KiB blocks of the 4 waves tile the A region [0, WK*BM) and the B region exactly once each
    std::vector<int> cover(G::BUF, 0);
    for (int w = 0; w < 4; ++w) {
        for (int j = 0; j < G::IA; ++j) {
            const int a = G::dma_a(w, j);
            CHECK(a >= 0 && a + 256 <= WK * G::BM, "%s: A block of wave %d instr %d at %d leaves [0,%d)", name, w, j, a, WK * G::BM);
            for (int k = 0; k < 256 && a + k < G::BUF && a >= 0; ++k) cover[a + k]++;
        }
        for (int j = 0; j < G::IB; ++j) {
            const int b = G::dma_b(w, j);
            CHECK(b >= WK * G::BM && b + 256 <= G::BUF, "%s: B block of wave %d instr %d at %d leaves [%d,%d)", name, w, j, b, WK * G::BM, G::BUF);
            for (int k = 0; k < 256 && b + k < G::BUF && b >= 0; ++k) cover[b + k]++;
        }
    }
    for (int i = 0; i < G::BUF; ++i) CHECK(cover[i] == 1, "%s: LDS float %d is filled %d times per K-step", name, i, cover[i]);
    // fragment reads of every wave / lane / k-pair
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int kp = 0; kp < WK / 2; ++kp) {
                const int a = G::frag_a(w / WN, l, kp), b = G::frag_b(w % WN, l, kp);
                CHECK(a >= 0 && a + 1 < WK * G::BM, "%s: A fragment read at %d (wave %d lane %d kp %d)", name, a, w, l, kp);
                CHECK(b >= WK * G::BM && b + 32 < G::BUF, "%s: B fragment read at %d (wave %d lane %d kp %d)", name, b, w, l, kp);
            }
    std::printf("ok %s: table %d entries/half, buffer %d floats\n", name, G::TAB, G::BUF);
}

static void check_bf16() {
    using G = WgradBf16Geom;
    for (int ks = 0; ks < 4 * G::TB; ++ks) {
        std::set<int> seen;
        for (int w = 0; w < 4; ++w)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < G::IB; ++j) {
                    const int i = G::tab_index(ks, w, l, j);
                    CHECK(i >= 0 && i < G::TAB && i / G::WK == ks % G::TB, "bf16: table index %d (ks %d wave %d lane %d j %d)", i, ks, w, l, j);
                    seen.insert(i);
                }
        CHECK((int)seen.size() == G::WK, "bf16: K-step %d touches %d table entries", ks, (int)seen.size());
    }
    std::vector<int> rows(G::WK, 0);
    for (int w = 0; w < 4; ++w)
        for (int j = 0; j < G::IA; ++j) {
            const int r = G::dma_row(w, j);
            CHECK(r >= 0 && r + 4 <= G::WK, "bf16: DMA rows %d..%d of wave %d instr %d", r, r + 3, w, j);
            for (int k = 0; k < 4 && r + k < G::WK && r >= 0; ++k) rows[r + k]++;
        }
    for (int r = 0; r < G::WK; ++r) CHECK(rows[r] == 1, "bf16: image row %d filled %d times", r, rows[r]);
    // transposing reads: 8-byte aligned, inside the image, and the operand they assemble is the one the MFMA wants
    for (int w2 = 0; w2 < 2; ++w2)
        for (int t = 0; t < 2; ++t)
            for (int kh = 0; kh < 2; ++kh)
                for (int rd = 0; rd < 2; ++rd)
                    for (int l = 0; l < 64; ++l) {
                        const int a = G::tr_addr(w2, t, rd, kh, l);
                        CHECK(a >= 0 && a + 8 <= G::IMG && (a & 7) == 0, "bf16: transposing read at %d (sub-tile %d/%d kh %d rd %d lane %d)", a, w2, t, kh, rd, l);
                        CHECK(G::tr_addr(w2, t, rd, kh, l) == G::tr_addr(w2, t, rd, 0, l) + kh * 16 * G::ROWB, "bf16: the pixel half is not a constant offset");
                        // the lane supplies block row q, columns 4p..4p+3: logical position of its 8 bytes
                        const int row = a / G::ROWB, chunk = ((a % G::ROWB) / 16) ^ G::fx(row), col = 8 * chunk + ((a & 8) ? 4 : 0);
                        const int g = l >> 4, q = (l & 15) >> 2, pp = l & 3;
                        CHECK(row == 16 * kh + 8 * (g >> 1) + 4 * rd + q && col == w2 * 64 + t * 32 + 16 * (g & 1) + 4 * pp,
                              "bf16: lane %d supplies row %d col %d", l, row, col);
                    }
    std::printf("ok bf16: table %d entries/half, image %d bytes\n", G::TAB, G::IMG);
}

static void check_split_once() {
    using G = WgradSplitGeom;
    // stores: the 256 threads x 2 halves tile a plane image exactly once, 8 bytes each, and land where the reads look for them
    std::vector<int> cover(G::IMG, 0);
    for (int tid = 0; tid < 256; ++tid)
        for (int half = 0; half < 2; ++half) {
            const int a = G::wr_addr(tid, half);
            CHECK(a >= 0 && a + 8 <= G::IMG && (a & 7) == 0, "split-once: store at %d (thread %d half %d)", a, tid, half);
            for (int k = 0; k < 8 && a >= 0 && a + k < G::IMG; ++k) cover[a + k]++;
            const int row = a / G::ROWB, chunk16 = ((a % G::ROWB) / 16) ^ WgradBf16Geom::fx(row), col = 8 * chunk16 + ((a & 8) ? 4 : 0);
            CHECK(row == G::pixel(tid, half) && col == 4 * G::chunk(tid), "split-once: thread %d half %d stores row %d col %d", tid, half, row, col);
            const int i = G::tab_index(5, tid, half);
            CHECK(i >= 0 && i < G::TAB && i / G::WK == 5 % G::TB && i % G::WK == G::pixel(tid, half), "split-once: table index %d", i);
        }
    for (int i = 0; i < G::IMG; ++i) CHECK(cover[i] == 1, "split-once: image byte %d stored %d times", i, cover[i]);
    for (int w2 = 0; w2 < 2; ++w2)
        for (int t = 0; t < 2; ++t)
            for (int rd = 0; rd < 2; ++rd)
                for (int l = 0; l < 64; ++l) {
                    const int a = G::tr_addr(w2, t, rd, l);
                    CHECK(a >= 0 && a + 8 <= G::IMG && (a & 7) == 0, "split-once: transposing read at %d (sub-tile %d/%d rd %d lane %d)", a, w2, t, rd, l);
                }
    std::printf("ok split-once: image %d bytes, buffer %d bytes\n", G::IMG, G::BUF);
}

// conv_igemm_mf16.hip: operand reads of every lane group conflict-free, and the direct-to-LDS fill consistent with them
static void check_mf16() {
    using G = Mf16Geom;
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    for (int t = 0; t < 8; ++t)
        for (int g = 0; g < 4; ++g) {
            std::set<int> slots;
            for (int i = 0; i < 16; ++i) {
                const int a = G::read_addr(groups[g][i], t);
                CHECK(a >= 0 && a + 16 <= 128 * G::ROWB && a % 16 == 0, "mf16: read address %d of lane %d, tile %d", a, groups[g][i], t);
                slots.insert((a % 256) / 16);
            }
            CHECK((int)slots.size() == 16, "mf16: lane group %d of row block %d covers %d of the 16 slots of the bank row", g, t, (int)slots.size());
        }
    // a DMA instruction fills 16 rows x 64 bytes linearly: lane -> row lane >> 2, slot lane & 3; the chunk it fetches must be the one the
    // readers expect at that slot, in every 16-row block (16 does not change (row >> 3) & 1's pattern within a block)
    for (int blk = 0; blk < 8; ++blk)
        for (int l = 0; l < 64; ++l) {
            const int row = 16 * blk + (l >> 2);
            CHECK(G::slot(row, G::dma_chunk(l)) == (l & 3), "mf16: lane %d of block %d fetches chunk %d, which belongs at slot %d", l, blk,
                  G::dma_chunk(l), G::slot(row, G::dma_chunk(l)));
        }
    // and every row's four chunks occupy its four slots
    for (int row = 0; row < 128; ++row) {
        std::set<int> s4;
        for (int c = 0; c < 4; ++c) s4.insert(G::slot(row, c));
        CHECK((int)s4.size() == 4, "mf16: row %d places two chunks in one slot", row);
    }
    std::printf("ok mf16: weight planes of 64-byte rows, ds_read_b128 lane groups conflict-free\n");
}

// conv_igemm_mf16.hip, staged activations: the 16-lane groups of both operand reads conflict-free, the fill consistent with the reads
static void check_mf16_a() {
    using G = Mf16AGeom;
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    for (int sm = 0; sm < 2; ++sm)
        for (int h = 0; h < 2; ++h)
            for (int g = 0; g < 4; ++g) {
                std::set<int> slots;
                for (int i = 0; i < 16; ++i) {
                    const int a = G::read_addr(groups[g][i], sm, h);
                    CHECK(a >= 0 && a + 16 <= G::WAVE_BYTES && a % 16 == 0, "mf16 A: read address %d of lane %d", a, groups[g][i]);
                    slots.insert((a % 256) / 16);
                }
                CHECK((int)slots.size() == 16, "mf16 A: lane group %d (block %d, read %d) covers %d of the 16 slots", g, sm, h, (int)slots.size());
            }
    // every (row, chunk) is fetched exactly once, into the slot its reader expects; a lane reads chunks 2g, 2g + 1 of its row
    std::set<int> filled;
    for (int j = 0; j < 4; ++j)
        for (int l = 0; l < 64; ++l) {
            const int row = G::dma_row(l, j), c = G::dma_chunk(l, j);
            CHECK(row >= 0 && row < G::ROWS && c >= 0 && c < 8, "mf16 A: instruction %d lane %d fetches row %d chunk %d", j, l, row, c);
            CHECK((c ^ G::fz(row)) == (l & 7), "mf16 A: instruction %d lane %d: chunk %d of row %d belongs at slot %d", j, l, c, row, c ^ G::fz(row));
            CHECK(G::ROWB * row + 16 * (l & 7) == 1024 * j + 16 * l, "mf16 A: instruction %d lane %d does not land linearly", j, l);
            filled.insert(row * 8 + c);
        }
    CHECK((int)filled.size() == 32 * 8, "mf16 A: %d of 256 (row, chunk) pairs fetched", (int)filled.size());
    for (int sm = 0; sm < 2; ++sm)
        for (int l = 0; l < 64; ++l)
            for (int h = 0; h < 2; ++h) {
                const int a = G::read_addr(l, sm, h), row = a / G::ROWB, c = ((a % G::ROWB) / 16) ^ G::fz(row);
                CHECK(row == 16 * sm + (l & 15) && c == 2 * (l >> 4) + h, "mf16 A: lane %d reads row %d chunk %d", l, row, c);
            }
    std::printf("ok mf16 A: staged activation rows of 128 bytes, ds_read_b128 lane groups conflict-free, row-coalesced fill\n");
}

int main() {
    check_fp32<1, 4, 16>("fp32 64x256");       // the three instances rn_conv_wgrad_batched launches (conv_wgrad.hip)
    check_fp32<4, 1, 16>("fp32 256x64");
    check_fp32<2, 2, 16>("fp32 128x128");
    check_fp32<2, 2, 32>("fp32 128x128 / 32-pixel steps");
    check_bf16();
    check_split_once();
    check_mf16();
    check_mf16_a();
    if (fails) std::printf("%d violation(s)\n", fails);
    return fails ? 1 : 0;
}
