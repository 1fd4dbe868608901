// tile_perm.hpp — host side of the trunk kernel's edge tiles (trunk.hpp TrunkArgs::perm / boff, conv_taps_static): where the boards of a
// workgroup's tile sit in the LDS image and which image row each MFMA row computes.  Plain C++ (no HIP): resnet.hip builds the tables at load
// time, tests/emu exposes the functions to the CPU test-suite (tests/test_tile_perm.py, tools/lds_bank_sim.py).
//
// The tile is `rows` MFMA rows = rows / 16 MFMA tiles; a wave computes `wave_rows` consecutive MFMA rows, and the kernel lets a wave's FIRST
// `per_wave_row` MFMA tiles sit out a tap when all 16 of their cells read zero padding there.  So the cells of each board edge (y = 0, y = H - 1,
// x = 0, x = W - 1; topped up with padding rows) become whole MFMA tiles — each sits out the three taps that look across its edge — dealt out
// over the wave rows; every other row goes to one of the remaining tiles.
//
// Bank conflicts (round 3).  A ds_read_b128 serves the lanes of a 16-row MFMA tile in two groups of eight rows ({0-3, 12-15} and {4-11}), and
// with the image swizzle (trunk.hpp swz_slot<true>) eight rows hit eight different bank groups iff they differ in row & 7 — on EVERY tap, since
// a tap shifts all rows of a tile by the same offset.  A tile therefore reads conflict-free iff it holds exactly TWO rows of every residue mod 8.
// The round-2 builder picked greedily and left ~7 clashes in the 128-row tile and ~10 in the 96-row tile: SQ_LDS_BANK_CONFLICT 9.6 M -> 35.6 M
// cycles per launch (tools/lds_bank_sim.py reproduces both figures from the bank rule).  With the boards at their natural image rows (b * H * W) no
// assignment is clash-free — the cells of the four edges do not split two-per-residue (an exact search says at least 5 resp. 10 clashes) — but
// WHERE a board sits in the image is free as long as its cells stay contiguous (neighbours are row +- 1, +- W): shifting the third Connect4
// board by one row (offsets 0, 42, 85) resp. the second by two (0, 44) makes a perfect split possible.  So: board offsets are searched
// (tile_layout), and for given offsets the rows are dealt to (tile, residue) buckets of capacity two by an exact bipartite matching.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace gaz {

struct TileLayout {
    std::vector<uint8_t> perm;      // [MFMA row] -> image row, a bijection on [0, rows); empty = no layout
    std::vector<int> boff;          // image row of cell 0 of board b
    int clashes = 0;                // sum over tiles and residues of max(0, rows of that residue - 2): 0 = every fragment read conflict-free
};

// (board, cell) of image row r, or board = -1 for a padding row
inline void tile_locate(const std::vector<int>& boff, int HW, int r, int& board, int& cell) {
    board = -1; cell = 0;
    for (size_t b = 0; b < boff.size(); ++b) if (r >= boff[b] && r < boff[b] + HW) { board = (int)b; cell = r - boff[b]; }
}

// taps (bit q = tap q of the 3x3 stencil) on which all 16 cells of MFMA tile `tile` of a permuted workgroup tile read zero padding
inline unsigned tile_sitout(const std::vector<uint8_t>& perm, const std::vector<int>& boff, int H, int W, int tile) {
    unsigned m = 0x1FFu;
    for (int i = 0; i < 16; ++i) {
        int b, cell; tile_locate(boff, H * W, perm[tile * 16 + i], b, cell);
        if (b < 0) continue;                        // padding row: reads zeros on every tap
        const int y = cell / W, x = cell % W;
        for (int q = 0; q < 9; ++q) { const int dy = q / 3 - 1, dx = q % 3 - 1; if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) m &= ~(1u << q); }
    }
    return m;
}

// rows -> tiles for GIVEN board offsets.  exact = true: only a clash-free assignment counts (empty result otherwise).
inline TileLayout tile_perm_at(int H, int W, const std::vector<int>& boff, int rows, int wave_rows, int per_wave_row, bool exact) {
    TileLayout L; L.boff = boff;
    const int HW = H * W, boards = (int)boff.size(), n_wr = rows / wave_rows, nt = rows / 16, max_special = std::min(per_wave_row * n_wr, 4);
    if (rows > 256 || rows % 16 || wave_rows % 16 || rows % wave_rows || wave_rows < 32 || boards < 1) return L;
    for (int b = 0; b < boards; ++b) if (boff[b] < 0 || boff[b] + HW > rows || (b && boff[b] < boff[b - 1] + HW)) return L;
    auto group_of = [&](int r, int g) {             // is image row r a cell of edge g (0: y = 0, 1: y = H - 1, 2: x = 0, 3: x = W - 1)?
        int b, cell; tile_locate(boff, HW, r, b, cell);
        if (b < 0) return false;
        const int y = cell / W, x = cell % W;
        return g == 0 ? y == 0 : g == 1 ? y == H - 1 : g == 2 ? x == 0 : x == W - 1;
    };
    auto is_pad = [&](int r) { int b, cell; tile_locate(boff, HW, r, b, cell); return b < 0; };
    // special tile k (edge k) -> wave row k % n_wr, MFMA tile k / n_wr of it
    std::vector<int> edge_of_tile(nt, -1);
    for (int k = 0; k < max_special; ++k) edge_of_tile[(k % n_wr) * (wave_rows / 16) + k / n_wr] = k;
    auto allowed = [&](int r, int t) { return edge_of_tile[t] < 0 || is_pad(r) || group_of(r, edge_of_tile[t]); };
    // exact part: rows <-> 2 slots per (tile, residue), Kuhn's augmenting paths.  Edge cells first (they have the fewest choices).
    std::vector<int> slot_row(nt * 16, -1), row_slot(rows, -1);      // slot = (t * 8 + residue) * 2 + {0, 1}
    std::vector<char> seen;
    struct Aug {
        std::vector<int>& slot_row; std::vector<int>& row_slot; std::vector<char>& seen; int nt;
        bool run(int r, const std::vector<std::vector<int>>& tiles_of) {
            for (int t : tiles_of[r])
                for (int h = 0; h < 2; ++h) {
                    const int s = (t * 8 + (r & 7)) * 2 + h;
                    if (seen[s]) continue;
                    seen[s] = 1;
                    if (slot_row[s] < 0 || run(slot_row[s], tiles_of)) { slot_row[s] = r; row_slot[r] = s; return true; }
                }
            return false;
        }
    };
    std::vector<std::vector<int>> tiles_of(rows);
    for (int r = 0; r < rows; ++r) {
        // preference order: a row that belongs to an edge tries that edge's tile first; other rows try the plain tiles in order
        for (int t = 0; t < nt; ++t) if (edge_of_tile[t] >= 0 && !is_pad(r) && group_of(r, edge_of_tile[t])) tiles_of[r].push_back(t);
        for (int t = 0; t < nt; ++t) if (edge_of_tile[t] < 0) tiles_of[r].push_back(t);
        for (int t = 0; t < nt; ++t) if (edge_of_tile[t] >= 0 && is_pad(r)) tiles_of[r].push_back(t);
    }
    std::vector<int> order(rows);
    for (int r = 0; r < rows; ++r) order[r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tiles_of[a].size() < tiles_of[b].size(); });
    // a special tile must end up FULL of its own edge's cells / padding: the matching is over all rows and all slots, so a perfect matching
    // fills every tile with exactly two rows per residue and respects `allowed` by construction
    int matched = 0;
    Aug aug{slot_row, row_slot, seen, nt};
    for (int r : order) { seen.assign(nt * 16, 0); if (aug.run(r, tiles_of)) ++matched; }
    std::vector<std::vector<int>> tiles(nt);
    if (matched == rows) {
        for (int s = 0; s < nt * 16; ++s) tiles[s / 16].push_back(slot_row[s]);
    } else {
        if (exact) return L;
        // no clash-free assignment at these offsets: keep what was matched, top the tiles up with the rest (clashes, counted below)
        std::vector<int> rest;
        for (int s = 0; s < nt * 16; ++s) if (slot_row[s] >= 0) tiles[s / 16].push_back(slot_row[s]);
        for (int r = 0; r < rows; ++r) if (row_slot[r] < 0) rest.push_back(r);
        for (int pass = 0; pass < 2; ++pass)                           // special tiles first: they may only take their own cells or padding
            for (int t = 0; t < nt; ++t) {
                if ((edge_of_tile[t] >= 0) != (pass == 0)) continue;
                for (size_t i = 0; i < rest.size() && (int)tiles[t].size() < 16;)
                    if (allowed(rest[i], t)) { tiles[t].push_back(rest[i]); rest.erase(rest.begin() + i); } else ++i;
            }
        for (int t = 0; t < nt; ++t) if ((int)tiles[t].size() != 16) return L;      // an edge without enough cells: no layout
    }
    // inside a tile: lanes {0-3, 12-15} and {4-11} each get one row of every residue where the tile has two
    L.perm.assign(rows, 0);
    const int P1[8] = {0, 1, 2, 3, 12, 13, 14, 15}, P2[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    for (int t = 0; t < nt; ++t) {
        std::sort(tiles[t].begin(), tiles[t].end());
        std::vector<int> bucket[8], g1, g2;
        for (int r : tiles[t]) bucket[r & 7].push_back(r);
        for (int res = 0; res < 8; ++res) L.clashes += std::max(0, (int)bucket[res].size() - 2);
        for (int res = 0; res < 8; ++res) if (!bucket[res].empty()) { g1.push_back(bucket[res].front()); bucket[res].erase(bucket[res].begin()); }
        for (int res = 0; res < 8; ++res) if (!bucket[res].empty() && g2.size() < 8) { g2.push_back(bucket[res].front()); bucket[res].erase(bucket[res].begin()); }
        for (int res = 0; res < 8; ++res) for (int r : bucket[res]) (g1.size() < 8 ? g1 : g2).push_back(r);
        while (g1.size() > 8) { g2.push_back(g1.back()); g1.pop_back(); }
        while (g2.size() > 8) { g1.push_back(g2.back()); g2.pop_back(); }
        for (int i = 0; i < 8; ++i) { L.perm[t * 16 + P1[i]] = (uint8_t)g1[i]; L.perm[t * 16 + P2[i]] = (uint8_t)g2[i]; }
    }
    std::vector<char> hit(rows, 0);
    for (int r = 0; r < rows; ++r) { if (hit[L.perm[r]]) { L.perm.clear(); return L; } hit[L.perm[r]] = 1; }      // must be a bijection
    return L;
}

// Board offsets + permutation: the first offsets (natural ones first, then by total shift) with a clash-free assignment; if there are none, the
// natural offsets with the fewest clashes the matching leaves.  want[] (optional, one mask per MFMA tile, 0 = any): sit-out masks the caller's kernel
// variant assumes — an assignment that does not deliver them is skipped.
inline TileLayout tile_layout(int H, int W, int boards, int rows, int wave_rows, int per_wave_row = 2, const unsigned* want = nullptr) {
    const int HW = H * W, slack = rows - boards * HW;
    TileLayout best;
    if (slack < 0 || boards < 1) return best;
    auto good = [&](const TileLayout& L) {
        if (L.perm.empty()) return false;
        for (int t = 0; want && t < rows / 16; ++t) if (want[t] && (tile_sitout(L.perm, L.boff, H, W, t) & want[t]) != want[t]) return false;
        return true;
    };
    // gaps g[b] >= 0 in front of board b with sum <= slack, enumerated by total shift
    std::vector<int> g(boards, 0);
    for (int total = 0; boards <= 4 && total <= slack * boards; ++total) {      // (more boards than the kernel's packed offsets hold: natural offsets only)
        // all compositions of `total` weighted shift: simple odometer over g with sum(g) <= slack, filtered by sum of offsets' shifts == total
        std::vector<int> gg(boards, 0);
        while (true) {
            int sum = 0, shift = 0, acc = 0;
            for (int b = 0; b < boards; ++b) { sum += gg[b]; acc += gg[b]; shift += acc; }
            if (sum <= slack && shift == total) {
                std::vector<int> boff(boards); int o = 0;
                for (int b = 0; b < boards; ++b) { o += gg[b]; boff[b] = o; o += HW; }
                TileLayout L = tile_perm_at(H, W, boff, rows, wave_rows, per_wave_row, true);
                if (good(L)) return L;
            }
            int i = 0;
            while (i < boards && ++gg[i] > slack) gg[i++] = 0;
            if (i == boards) break;
        }
    }
    std::vector<int> nat(boards);
    for (int b = 0; b < boards; ++b) nat[b] = b * HW;
    best = tile_perm_at(H, W, nat, rows, wave_rows, per_wave_row, false);
    if (!good(best)) best.perm.clear();
    return best;
}

}  // namespace gaz
