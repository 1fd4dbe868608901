// tile_perm.hpp — host side of the trunk kernel's edge tiles (trunk.hpp TrunkArgs::perm, conv_taps_static): which cell of a workgroup's
// tile each MFMA row computes.  Plain C++ (no HIP): resnet.hip builds the tables at load time, tests/emu exposes the functions to the CPU
// test-suite (tests/test_tile_perm.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace gaz {

// [MFMA row of a workgroup's tile] -> image row (cell of the tile; rows >= boards * H * W are padding), for TrunkArgs::perm.  The tile is
// `rows` MFMA rows = rows / 16 MFMA tiles; a wave computes `wave_rows` consecutive MFMA rows, and the kernel can let a wave's FIRST `per_wave_row` MFMA
// tiles sit out a tap when all 16 of their cells read zero padding there.  So: the cells of each board edge (y = 0, y = H - 1, x = 0,
// x = W - 1; topped up with padding rows) become whole MFMA tiles — each sits out the three taps that look across its edge — dealt out
// over the waves; leftover padding rows make all-padding tiles (they sit out every tap); every other cell keeps its natural order.  Inside
// a tile the rows are ordered so that the two groups of eight lanes a ds_read_b128 serves together ({0-3, 12-15} and {4-11}) hit eight
// different row & 7 (the image swizzle's conflict-free condition).  Connect4, three boards in 128 rows: four edge tiles, 30 of 36 tile-taps
// per wave left (-16.7 % MFMAs); two boards in 96 rows: two edge tiles (-11 %).  (A Gomoku board in 256 rows would give four edge tiles, one per
// wave row, -8.3 %: that is four instances of the kernel's block loop, which spill — see trunk.hpp — so the Gomoku launch keeps the natural order.)
inline std::vector<uint8_t> tile_perm(int H, int W, int boards, int rows, int wave_rows, int per_wave_row = 2) {
    const int HW = H * W, cells = boards * HW, n_wr = rows / wave_rows, max_special = per_wave_row * n_wr;
    std::vector<uint8_t> none;
    if (rows > 256 || rows % 16 || wave_rows % 16 || rows % wave_rows || cells > rows || wave_rows < 32) return none;
    std::vector<char> used(rows, 0);
    std::vector<int> pads;
    for (int r = rows - 1; r >= cells; --r) pads.push_back(r);      // taken from the back: lowest padding row first
    auto edges_of = [&](int c) { const int cell = c % HW, y = cell / W, x = cell % W; return (y == 0) + (y == H - 1) + (x == 0) + (x == W - 1); };
    auto in_group = [&](int c, int g) { const int cell = c % HW, y = cell / W, x = cell % W; return g == 0 ? y == 0 : g == 1 ? y == H - 1 : g == 2 ? x == 0 : x == W - 1; };
    // a tile reads without LDS bank conflicts when each of its two groups of eight lanes sees eight different row & 7, i.e. when it holds at
    // most two rows of every residue: pick with that in mind (every residue has rows / 8 rows, so a perfect split exists for the whole tile)
    std::vector<std::vector<int>> special;
    auto pick16 = [&](const std::vector<int>& cand, std::vector<int>& t) {       // up to 16 - t.size() more rows from cand, at most two per residue first
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r : t) cnt[r & 7]++;
        for (int pass = 0; pass < 2; ++pass)
            for (int c : cand) {
                if ((int)t.size() >= 16) return;
                if (std::find(t.begin(), t.end(), c) != t.end() || used[c]) continue;
                if (pass == 0 && cnt[c & 7] >= 2) continue;
                t.push_back(c); cnt[c & 7]++;
            }
    };
    for (int g = 0; g < 4 && (int)special.size() < max_special; ++g) {
        std::vector<int> avail;
        for (int pass = 1; pass <= 2; ++pass)                       // cells of this edge only first, corners (shared with another edge) last
            for (int c = 0; c < cells; ++c) if (!used[c] && in_group(c, g) && (edges_of(c) == 1) == (pass == 1)) avail.push_back(c);
        if (avail.empty() || 16 - std::min<int>(16, (int)avail.size()) > (int)pads.size()) continue;
        std::vector<int> t;
        pick16(avail, t);
        while ((int)t.size() < 16) { t.push_back(pads.back()); pads.pop_back(); }
        for (int r : t) used[r] = 1;
        special.push_back(t);
    }
    while ((int)special.size() < max_special && pads.size() >= 16) {
        std::vector<int> t;
        for (int i = 0; i < 16; ++i) { t.push_back(pads.back()); pads.pop_back(); }
        for (int r : t) used[r] = 1;
        special.push_back(t);
    }
    // special tile k -> wave-row k % n_wr, MFMA tile k / n_wr of it; the other tiles share out the remaining rows, two per residue each
    std::vector<std::vector<int>> tiles(rows / 16);
    std::vector<char> is_special(rows / 16, 0);
    for (size_t k = 0; k < special.size(); ++k) { const size_t ti = (k % n_wr) * (wave_rows / 16) + k / n_wr; tiles[ti] = special[k]; is_special[ti] = 1; }
    std::vector<int> rest;
    for (int r = 0; r < rows; ++r) if (!used[r]) rest.push_back(r);
    for (size_t ti = 0; ti < tiles.size(); ++ti) {
        if (is_special[ti]) continue;
        pick16(rest, tiles[ti]);
        for (int r : tiles[ti]) used[r] = 1;
    }
    std::vector<uint8_t> perm(rows);
    const int P1[8] = {0, 1, 2, 3, 12, 13, 14, 15}, P2[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    for (size_t ti = 0; ti < tiles.size(); ++ti) {
        std::vector<int> bucket[8], g1, g2;
        for (int r : tiles[ti]) bucket[r & 7].push_back(r);
        for (int res = 0; res < 8; ++res) if (!bucket[res].empty()) { g1.push_back(bucket[res].back()); bucket[res].pop_back(); }
        for (int res = 0; res < 8; ++res) if (!bucket[res].empty() && g2.size() < 8) { g2.push_back(bucket[res].back()); bucket[res].pop_back(); }
        for (int res = 0; res < 8; ++res) for (int r : bucket[res]) (g1.size() < 8 ? g1 : g2).push_back(r);
        while (g1.size() > 8) { g2.push_back(g1.back()); g1.pop_back(); }
        while (g2.size() > 8) { g1.push_back(g2.back()); g2.pop_back(); }
        for (int i = 0; i < 8; ++i) { perm[ti * 16 + P1[i]] = (uint8_t)g1[i]; perm[ti * 16 + P2[i]] = (uint8_t)g2[i]; }
    }
    std::vector<char> seen(rows, 0);
    for (int r = 0; r < rows; ++r) { if (seen[perm[r]]) return none; seen[perm[r]] = 1; }       // must be a bijection
    return perm;
}

// taps (bit q = tap q of the 3x3 stencil) on which all 16 cells of MFMA tile `tile` of a permuted workgroup tile read zero padding
inline unsigned tile_sitout(const std::vector<uint8_t>& perm, int H, int W, int boards, int tile) {
    unsigned m = 0x1FFu;
    for (int i = 0; i < 16; ++i) {
        const int r = perm[tile * 16 + i];
        if (r >= boards * H * W) continue;          // padding row: reads zeros on every tap
        const int cell = r % (H * W), y = cell / W, x = cell % W;
        for (int q = 0; q < 9; ++q) { const int dy = q / 3 - 1, dx = q % 3 - 1; if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) m &= ~(1u << q); }
    }
    return m;
}

}  // namespace gaz
