// gumbel_core.hpp — Gumbel AlphaZero search (sequential halving at the root, completed-Q deterministic selection
// below it) and the use_gumbel branch of the self-play loop, one wavefront per game.
//
// Reference path replaced (file:line under /root/reference):
//   stablemax / softmax / q_transform / rescale_q / sigma / compute_v_mix / compute_pi   MCTS_Gumbel.py:75-148   -> compute_pi()
//   MCTS_Gumbel.sequential_halving                                                    MCTS_Gumbel.py:212-224  -> g_halving()
//   MCTS_Gumbel.deterministic_selection / select                                      MCTS_Gumbel.py:226-260  -> g_select()
//   MCTS_Gumbel.get_terminal_actions_fn (unsorted)                                    MCTS_Gumbel.py:281-318  -> terminal_probe_unsorted()
//   MCTS_Gumbel.create_expand_root / _expand / _expand_with_terminal_actions          MCTS_Gumbel.py:320-528  -> g_root_pre/post, g_expand_pre/post
//   MCTS_Gumbel._back_propagate                                                       MCTS_Gumbel.py:530-546  -> backup() (shared with PUCT)
//   MCTS_Gumbel.run                                                                   MCTS_Gumbel.py:562-679  -> phases of g_game_step
//   Self_Play.create_MCTS_Gumbel / play (use_gumbel)                                  Self_Play.py:58-69,108-153 (fresh tree every move)
//
// Numerics follow numpy-without-Numba (how the fixtures were recorded): float32 statistics, float64 softmax with the
// det:: exp, numpy's pairwise summation order, stable ascending argsort.  A node keeps its children in LEGAL-ACTION
// order (no prior sort), with raw logits in P[] and the evaluator value of each expanded child in RAW[].
#pragma once
#include "puct_core.hpp"

namespace gaz {

template <class G> struct GumbelState {          // per game, lives across launches of one MCTS_Gumbel.run
    int32_t m_eff, phase, n_top, cand, stage, sims_left, vpc, cur_iter, pend_counts;
    int32_t iter_limit, pad_;  // this move's iteration_limit: run_iterations, or 3 x legal moves under a time limit (DevParams::move_time_ticks)
    float top_logits[G::APAD];
    float top_mean[G::APAD];
    uint8_t top_ids[G::APAD];
};

template <class G> GAZ_DEV float* node_raw(const NodeRef<G>& nd) {
    return reinterpret_cast<float*>(nd.p + NodeLayout<G>::SIZE);      // RAW[APAD] is appended after the PUCT record
}
template <class G> constexpr int gumbel_node_bytes() { return NodeLayout<G>::SIZE + 4 * G::APAD; }

constexpr float F32_EPS = 1.1920928955078125e-07f;

// terminal moves in order of appearance (MCTS_Gumbel.py:301-318)
template <class G> GAZ_DEV int terminal_probe_unsorted(const int8_t* board, const uint8_t* legal, int n_legal, int player,
                                                       uint8_t* tact, uint8_t* twin, bool& any_win, bool fast_find_win = false) {
    if (fast_find_win && first_win_only<G>(board, legal, n_legal, player, tact, twin)) { any_win = true; return 1; }
    const int empties = G::DRAWS ? count_empty<G>(board) : 0;
    int nt = 0; uint64_t anyw = 0;
    for (int base = 0; base < n_legal; base += G::TEAM) {
        int i = base + tlane<G>();
        bool win = false, hit = false;
        if (i < n_legal) {
            win = wins_after<G>(board, landing_cell<G>(board, legal[i]), player);
            hit = win || (G::DRAWS && empties == 1);
        }
        uint64_t m = tballot<G>(hit);
        anyw |= tballot<G>(win);
        if (hit) { int pos = nt + popcll(m & ((1ull << tlane<G>()) - 1ull)); tact[pos] = legal[i]; twin[pos] = win ? 1 : 0; }
        nt += popcll(m);
    }
    any_win = anyw != 0;
    wave_sync();
    return nt;
}

// One coalesced read of a node record (header + N / W / logits / child / action blocks, and RAW[]) into LDS: a level of the
// descent then costs ONE dependent HBM round trip instead of one per block touched (three cache lines, read several times).
template <class G> GAZ_DEV NodeRef<G> g_stage_node(const DevParams<G>& E, int g, int node, Scratch<G>& S) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(node_at(E, g, 0, node).p);
    constexpr int NW = (NodeLayout<G>::OFF_BOARD + 3) / 4, RW0 = NodeLayout<G>::SIZE / 4;
    wave_sync();
    for (int i = tlane<G>(); i < NW + G::APAD; i += G::TEAM) {
        const uint32_t v = src[i < NW ? i : RW0 + (i - NW)];
        if (i < NW) S.node[i] = v; else reinterpret_cast<uint32_t*>(S.raw)[i - NW] = v;
    }
    wave_sync();
    return NodeRef<G>{reinterpret_cast<uint8_t*>(S.node)};
}

// softmax in float64 over S.gam[0..n): x <- exp(x - max) / np.sum(...)   (MCTS_Gumbel.py:82-88)
template <class G> GAZ_DEV void softmax_inplace(Scratch<G>& S, int n) {
    double mx = S.gam[0];                                                           // uniform
    if (n <= 8) {                                                                   // small node: the reads in one batch
        double gv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) gv[i] = S.gam[i];
#pragma unroll
        for (int i = 1; i < 8; ++i) if (i < n && gv[i] > mx) mx = gv[i];
    } else {
        for (int i = 1; i < n; ++i) { double v = S.gam[i]; if (v > mx) mx = v; }
    }
    wave_sync();
    const double c = -mx;
    for (int i = tlane<G>(); i < n; i += G::TEAM) S.gam[i] = det::dexp(S.gam[i] + c);
    wave_sync();
    const double s = det::np_pairwise_sum<double>(S.gam, n);
    wave_sync();
    for (int i = tlane<G>(); i < n; i += G::TEAM) S.gam[i] = S.gam[i] / s;
    wave_sync();
}

// compute_pi(use_softmax=True) for node nd (MCTS_Gumbel.py:126-141 with :113-124, :99-103, :106-110).  Result f32 in S.pri.
#ifndef GAZ_HOST_EMU
// value of TEAM lane `i` (a constant after unrolling, i < 8) in every lane of the team.  Whole-wave teams: v_readlane_b32 into a
// scalar register (a few cycles; __shfl would go through the LDS crossbar, ~100 cycles of latency each, and these reductions are
// chains of them).  16-lane teams: one DPP move with row_newbcast:i — lane i of EVERY 16-lane row to all lanes of its row, i.e. all
// four games of the wave at once.
template <int I> GAZ_DEV int row_bcast_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + I, 0xF, 0xF, false); }
template <class G> GAZ_DEV uint32_t tlane_val(uint32_t v, int i) {
    if (G::TEAM >= 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, i);
    switch (i) {
        case 0: return (uint32_t)row_bcast_i32<0>((int)v); case 1: return (uint32_t)row_bcast_i32<1>((int)v);
        case 2: return (uint32_t)row_bcast_i32<2>((int)v); case 3: return (uint32_t)row_bcast_i32<3>((int)v);
        case 4: return (uint32_t)row_bcast_i32<4>((int)v); case 5: return (uint32_t)row_bcast_i32<5>((int)v);
        case 6: return (uint32_t)row_bcast_i32<6>((int)v); default: return (uint32_t)row_bcast_i32<7>((int)v);
    }
}
template <class G> GAZ_DEV float tlane_val(float v, int i) { return __uint_as_float(tlane_val<G>(__float_as_uint(v), i)); }
template <class G> GAZ_DEV double tlane_val(double v, int i) {
    const uint64_t b = det::d2bits(v);
    const uint32_t lo = tlane_val<G>((uint32_t)b, i), hi = tlane_val<G>((uint32_t)(b >> 32), i);
    return det::bits2d(((uint64_t)hi << 32) | lo);
}
// sum of v over team lanes 0..n-1 in numpy's order for n < 8 (res = 0; res += a[i]), every lane gets the result
template <class G, class T> GAZ_DEV T seq_sum_lanes(T v, int n) {
    T res = (T)0;
#pragma unroll
    for (int i = 0; i < 7; ++i) { const T vi = tlane_val<G>(v, i); if (i < n) res = res + vi; }
    return res;
}
template <class G, class T> GAZ_DEV T max_lanes(T v, int n) {      // max over team lanes 0..n-1 (n >= 1), starting from lane 0 like the loop it replaces
    T m = tlane_val<G>(v, 0);
#pragma unroll
    for (int i = 1; i < 7; ++i) { const T vi = tlane_val<G>(v, i); if (i < n && vi > m) m = vi; }
    return m;
}
template <class G, class T> GAZ_DEV T min_lanes(T v, int n) {
    T m = tlane_val<G>(v, 0);
#pragma unroll
    for (int i = 1; i < 7; ++i) { const T vi = tlane_val<G>(v, i); if (i < n && vi < m) m = vi; }
    return m;
}

// compute_pi for nodes with fewer than 8 children (every Connect4 node): child i lives in lane i and the fourteen
// LDS write -> fence -> read phases of the general version become register shuffles; same operations in the same order.
template <class G> GAZ_DEV void compute_pi_small(const DevParams<G>& E, const NodeRef<G>& nd, const float* RAW, int n, Scratch<G>& S,
                                                 uint32_t& N_b_out, uint64_t& sumv_out, bool stable) {
    const int i = tlane<G>();
    const bool on = i < n;
    const uint32_t Ni = on ? nd.N()[i] : 0u;
    const float Wi = on ? nd.W()[i] : 0.0f, Li = on ? nd.P()[i] : 0.0f, RAWi = on ? RAW[i] : 0.0f;
    uint32_t nb = 0; uint64_t sumv = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) { const uint32_t v = tlane_val<G>(Ni, k); if (k < n) { if (v > nb) nb = v; sumv += v; } }
    double x, mx, e, ssum;
    float pri;
    if (stable) {                                           // stablemax(float32 logits)
        const float sf = Li >= 0.0f ? Li + 1.0f : 1.0f / ((1.0f - Li) + F32_EPS);
        pri = sf / seq_sum_lanes<G>(sf, n);
    } else {                                                // softmax #1 (float64)
        x = (double)Li;
        mx = max_lanes<G>(x, n);
        e = det::dexp(x + (-mx));
        ssum = seq_sum_lanes<G>(e, n);
        pri = (float)(e / ssum);
    }
    const float mean = Ni > 0 ? (float)((double)Wi / (double)Ni) : -1.0f;
    const float q = (mean - (-1.0f)) / 2.0f;
    const float sum_probs = seq_sum_lanes<G>(Ni > 0 ? pri : 0.0f, n);
    const float weighted_q = seq_sum_lanes<G>(Ni > 0 ? (pri * q) / sum_probs : 0.0f, n);
    const double wq = (double)weighted_q * (double)sumv;
    const float vmix = (float)(((double)RAWi + wq) / (double)(sumv + 1));
    const float cq = Ni > 0 ? q : vmix;                                                                // completed_q
    const float mn = min_lanes<G>(cq, n), mxq = max_lanes<G>(cq, n);
    const float den = (mxq - mn) > F32_EPS ? (mxq - mn) : F32_EPS;
    const double sg = (E.c_visit + (double)nb) * E.c_scale;
    const float r = (cq - mn) / den;
    x = (double)Li + sg * (double)r;
    if (stable) {                                           // stablemax in float64, left in S.gam
        const double sd = x >= 0.0 ? x + 1.0 : 1.0 / ((1.0 - x) + (double)F32_EPS);
        const double tot = seq_sum_lanes<G>(sd, n);
        if (on) S.gam[i] = sd / tot;
    } else {
        mx = max_lanes<G>(x, n);
        e = det::dexp(x + (-mx));
        ssum = seq_sum_lanes<G>(e, n);
        if (on) S.pri[i] = (float)(e / ssum);
    }
    wave_sync();
    N_b_out = nb; sumv_out = sumv;
}
#endif

// nd may be the LDS copy of the record (g_stage_node); RAW is passed separately because it is not contiguous with the header.
// stable = true: compute_pi(use_softmax=False) as numpy 2 evaluates it without Numba (MCTS_Gumbel.py:144-148): probs =
// stablemax(float32 logits); sigma is float64 ((c_visit + N_b) is a NumPy float64 scalar), so logits + sigma and the second
// stablemax are float64 and the result is NOT cast to float32: it is left in S.gam (S.pri is not written).
template <class G> GAZ_DEV void compute_pi(const DevParams<G>& E, const NodeRef<G>& nd, const float* RAW, int n, Scratch<G>& S, uint32_t& N_b_out,
                                           uint64_t& sumv_out, bool stable = false) {
#ifndef GAZ_HOST_EMU
    if (n < 8) { compute_pi_small<G>(E, nd, RAW, n, S, N_b_out, sumv_out, stable); return; }
#endif
    const uint32_t* N = nd.N(); const float* W = nd.W(); const float* L = nd.P();
    uint32_t nb = 0; uint64_t sumv = 0;                                                              // uniform
    if (n <= 8) {
        uint32_t nv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) nv[i] = N[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) if (i < n) { if (nv[i] > nb) nb = nv[i]; sumv += nv[i]; }
    } else {
        for (int i = 0; i < n; ++i) { uint32_t v = N[i]; if (v > nb) nb = v; sumv += v; }
    }
    if (stable) {                                           // stablemax(float32 logits), np.sum in float32
        for (int i = tlane<G>(); i < n; i += G::TEAM) { const float l = L[i]; S.spri[i] = l >= 0.0f ? l + 1.0f : 1.0f / ((1.0f - l) + F32_EPS); }
        wave_sync();
        const float ssum = det::np_pairwise_sum<float>(S.spri, n);
        wave_sync();
        for (int i = tlane<G>(); i < n; i += G::TEAM) S.gam[i] = (double)(S.spri[i] / ssum);
        wave_sync();
    } else {
        for (int i = tlane<G>(); i < n; i += G::TEAM) S.gam[i] = (double)L[i];
        wave_sync();
        softmax_inplace<G>(S, n);
    }
    // probs (f32) in S.pri; q in S.aux
    for (int i = tlane<G>(); i < n; i += G::TEAM) {
        S.pri[i] = (float)S.gam[i];
        const float mean = N[i] > 0 ? (float)((double)W[i] / (double)N[i]) : -1.0f;               // mean_values, q_transform
        S.aux[i] = (mean - (-1.0f)) / 2.0f;
    }
    wave_sync();
    for (int i = tlane<G>(); i < n; i += G::TEAM) S.spri[i] = N[i] > 0 ? S.pri[i] : 0.0f;
    wave_sync();
    const float sum_probs = det::np_pairwise_sum<float>(S.spri, n);
    wave_sync();
    for (int i = tlane<G>(); i < n; i += G::TEAM) S.spri[i] = N[i] > 0 ? (S.pri[i] * S.aux[i]) / sum_probs : 0.0f;
    wave_sync();
    const float weighted_q = det::np_pairwise_sum<float>(S.spri, n);
    wave_sync();
    const double wq = (double)weighted_q * (double)sumv;
    for (int i = tlane<G>(); i < n; i += G::TEAM) {
        const float vmix = (float)(((double)RAW[i] + wq) / (double)(sumv + 1));
        S.spri[i] = N[i] > 0 ? S.aux[i] : vmix;                                                   // completed_q
    }
    wave_sync();
    float mn = S.spri[0], mx = S.spri[0];                                                            // uniform
    if (n <= 8) {
        float sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sv[i] = S.spri[i];
#pragma unroll
        for (int i = 1; i < 8; ++i) if (i < n) { if (sv[i] < mn) mn = sv[i]; if (sv[i] > mx) mx = sv[i]; }
    } else {
        for (int i = 1; i < n; ++i) { float v = S.spri[i]; if (v < mn) mn = v; if (v > mx) mx = v; }
    }
    const float den = (mx - mn) > F32_EPS ? (mx - mn) : F32_EPS;
    const double sg = (E.c_visit + (double)nb) * E.c_scale;
    wave_sync();
    for (int i = tlane<G>(); i < n; i += G::TEAM) { const float r = (S.spri[i] - mn) / den; S.gam[i] = (double)L[i] + sg * (double)r; }
    wave_sync();
    if (stable) {                                           // stablemax in float64, result stays in S.gam
        for (int i = tlane<G>(); i < n; i += G::TEAM) { const double x = S.gam[i]; S.gam[i] = x >= 0.0 ? x + 1.0 : 1.0 / ((1.0 - x) + (double)F32_EPS); }
        wave_sync();
        const double ssum = det::np_pairwise_sum<double>(S.gam, n);
        wave_sync();
        for (int i = tlane<G>(); i < n; i += G::TEAM) S.gam[i] = S.gam[i] / ssum;
        wave_sync();
    } else {
        softmax_inplace<G>(S, n);
        for (int i = tlane<G>(); i < n; i += G::TEAM) S.pri[i] = (float)S.gam[i];
        wave_sync();
    }
    N_b_out = nb; sumv_out = sumv;
}

// deterministic_selection: argmax(pi - visits / (1 + sum(visits))) in float64, first maximum (MCTS_Gumbel.py:243)
template <class G> GAZ_DEV int g_det_select(const DevParams<G>& E, const NodeRef<G>& nd, const float* RAW, int n, Scratch<G>& S) {
    uint32_t nb; uint64_t sumv;
    const bool stable = E.g_stablemax != 0;
    compute_pi<G>(E, nd, RAW, n, S, nb, sumv, stable);
    double best = 0.0; int bi = 0x7fffffff;
    for (int i = tlane<G>(); i < n; i += G::TEAM) {
        double sc = (stable ? S.gam[i] : (double)S.pri[i]) - (double)nd.N()[i] / (double)(1 + sumv);
        if (bi == 0x7fffffff || sc > best) { best = sc; bi = i; }
    }
    team_argmax<G>(best, bi);
    wave_sync();
    return tuni<G>(bi);
}

// sequential_halving + the bookkeeping around it (MCTS_Gumbel.py:603-623).  Returns false when one candidate is left.
template <class G> GAZ_DEV bool g_halving(const DevParams<G>& E, GumbelState<G>& gu, const NodeRef<G>& r, int n_root, Scratch<G>& S) {
    const int m = gu.m_eff, phase = gu.phase, n_top = gu.n_top;
    double halved = (double)m / (double)(1 << (phase < 30 ? phase : 30)); if (halved < 1.0) halved = 1.0;
    uint32_t nb = 0; for (int i = 0; i < n_root; ++i) { uint32_t v = r.N()[i]; if (v > nb) nb = v; }
    const double sg = (E.c_visit + (double)nb) * E.c_scale;
    for (int i = tlane<G>(); i < n_top; i += G::TEAM) {
        double sc = (double)gu.top_logits[i];
        if (phase > 0) { const float qh = (gu.top_mean[i] - (-1.0f)) / 2.0f; sc = sc + sg * (double)qh; }
        S.gam[i] = sc;
    }
    wave_sync();
    int take = phase == 0 ? m : (int)halved; if (take > n_top) take = n_top;
    for (int i = tlane<G>(); i < n_top; i += G::TEAM) {          // stable ascending rank; keep the `take` largest in ascending order
        const double v = S.gam[i]; int rank = 0;
        for (int j = 0; j < n_top; ++j) { const double o = S.gam[j]; rank += (o < v) || (o == v && j < i); }
        const int pos = rank - (n_top - take);
        if (pos >= 0) { S.spri[pos] = gu.top_logits[i]; S.sact[pos] = gu.top_ids[i]; }
    }
    wave_sync();
    for (int i = tlane<G>(); i < take; i += G::TEAM) { gu.top_logits[i] = S.spri[i]; gu.top_ids[i] = S.sact[i]; }
    double lg2;                                              // np.log2(m): exact for powers of two
    if ((m & (m - 1)) == 0) { lg2 = 0.0; for (int mm = m; mm > 1; mm >>= 1) lg2 += 1.0; }
    else lg2 = det::dlog((double)m) / 0.6931471805599453;
    const int iters = tuni<G>(gu.iter_limit);
    int vpc = m > 1 ? (int)((double)iters / (lg2 * halved)) : iters; if (vpc < 1) vpc = 1;
    if (take == 2 || take == 3) { vpc = (iters - gu.cur_iter) / take; if (vpc < 1) vpc = 1; }
    wave_sync();
    if (tlane<G>() == 0) { gu.n_top = take; gu.vpc = vpc; gu.cand = 0; gu.stage = 0; }
    wave_sync();
    return take > 1;
}

// create_expand_root, first half
template <class G> GAZ_DEV bool g_root_pre(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, Scratch<G>& S) {
    if (tlane<G>() == 0) { ts.n_nodes = 0; ts.root = -1; ts.root_visits = 0; }
    wave_sync();
    copy_board<G>(S.board, gs.board);
    wave_sync();
    const int n_legal = build_legal<G>(S.board, S.legal);
    bool any_win;
    const int nt = terminal_probe_unsorted<G>(S.board, S.legal, n_legal, gs.next_player, S.tact, S.twin, any_win, E.fast_find_win != 0);
    uint8_t h3[3];
    for (int i = 0; i < 3; ++i) h3[i] = (gs.n_hist - 1 - i >= 0) ? gs.hist[gs.n_hist - 1 - i] : 0;
    const int idx = alloc_node(E, ts);
    if (idx < 0) return false;
    NodeRef<G> nd = node_at(E, g, 0, idx);
    if (tlane<G>() == 0) {
        NodeHdr h; memset(&h, 0, sizeof(h));
        h.parent = -1; h.player = (int8_t)(-gs.next_player); h.n_hist = (uint16_t)gs.n_hist;
        h.hist3[0] = h3[0]; h.hist3[1] = h3[1]; h.hist3[2] = h3[2]; h.action = h3[0];
        if (nt > 0) { h.n_actions = (uint8_t)nt; h.n_children = (uint8_t)nt; h.flags = NF_TERMINAL_PARENT; }
        *nd.hdr() = h; ts.root = idx;
    }
    copy_board<G>(nd.board(), S.board);
    if (nt > 0) {                                                              // MCTS_Gumbel.py:339-370
        for (int i = tlane<G>(); i < nt; i += G::TEAM) {
            const float mask = S.twin[i] ? 1.0f : 0.0f;
            nd.N()[i] = 1u; nd.W()[i] = any_win ? 1.0f : 0.0f; node_raw<G>(nd)[i] = mask;
            nd.P()[i] = any_win ? mask / (float)nt : 1.0f / (float)nt;
            nd.child()[i] = S.twin[i] ? CHILD_LEAF_WIN : CHILD_LEAF_DRAW; nd.act()[i] = S.tact[i];
        }
        if (tlane<G>() == 0) ts.root_visits = (uint64_t)nt;
        wave_sync();
        return false;
    }
    encode_input<G>(S.board, -gs.next_player, h3, gs.n_hist, E.nn_in + (size_t)g * (G::HW * G::C), E.done_flag != nullptr);
    wave_sync();
    return true;
}

// children of a freshly evaluated node: raw logits of the legal actions in legal order (normalize=False), all unexpanded
// `fresh`: S.board already is this node's board (same launch as g_expand_pre / g_root_pre: evaluation-cache hit)
template <class G> GAZ_DEV void g_write_children(const DevParams<G>& E, int g, const NodeRef<G>& nd, Scratch<G>& S, const float* policy,
                                                 bool fresh = false) {
    if (!fresh) { copy_board<G>(S.board, nd.board()); wave_sync(); }
    const int n_legal = build_legal<G>(S.board, S.legal);
    for (int i = tlane<G>(); i < n_legal; i += G::TEAM) {
        nd.N()[i] = 0u; nd.W()[i] = 0.0f; node_raw<G>(nd)[i] = 0.0f; nd.P()[i] = policy[S.legal[i]];
        nd.child()[i] = CHILD_NONE; nd.act()[i] = S.legal[i];
    }
    if (tlane<G>() == 0) { nd.hdr()->n_actions = (uint8_t)n_legal; nd.hdr()->n_children = 0; }
    wave_sync();
}

// _expand of child `index` of `node`; path[0..depth) leads to node.  true = evaluation pending
// `staged`: the descent left this node's header + child blocks in S.node (saves two dependent round trips)
template <class G> GAZ_DEV bool g_expand_pre(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, Scratch<G>& S,
                                             int node, int index, int depth, bool staged = false) {
    NodeRef<G> pn = node_at(E, g, 0, node);
    const NodeRef<G> ps = staged ? NodeRef<G>{reinterpret_cast<uint8_t*>(S.node)} : pn;
    const NodeHdr ph = *ps.hdr();
    const int action = tuni<G>((int)ps.act()[index]);
    const int mover = -(int)tuni<G>((int)ph.player);
    copy_board<G>(S.board, pn.board());
    wave_sync();
    const int cell = landing_cell<G>(S.board, action);
    wave_sync();
    if (tlane<G>() == 0) S.board[cell] = (int8_t)mover;
    wave_sync();
    const int n_legal = build_legal<G>(S.board, S.legal);
    bool any_win;
    const int nt = terminal_probe_unsorted<G>(S.board, S.legal, n_legal, -mover, S.tact, S.twin, any_win, E.fast_find_win != 0);
    const int idx = alloc_node(E, ts);
    if (idx < 0) return false;
    NodeRef<G> nd = node_at(E, g, 0, idx);
    if (tlane<G>() == 0) {
        NodeHdr h; memset(&h, 0, sizeof(h));
        h.parent = node; h.slot = (int16_t)index; h.player = (int8_t)mover; h.n_hist = (uint16_t)(ph.n_hist + 1);
        h.hist3[0] = (uint8_t)action; h.hist3[1] = ph.hist3[0]; h.hist3[2] = ph.hist3[1]; h.action = (uint8_t)action;
        if (nt > 0) { h.n_actions = (uint8_t)nt; h.n_children = (uint8_t)nt; h.flags = NF_TERMINAL_PARENT; }
        *nd.hdr() = h;
    }
    S.path[depth].node = node; S.path[depth].slot = index;
    copy_board<G>(nd.board(), S.board);
    wave_sync();
    if (nt > 0) {                                                              // MCTS_Gumbel.py:391-453: visits / values stay 0
        for (int i = tlane<G>(); i < nt; i += G::TEAM) {
            const float mask = S.twin[i] ? 1.0f : 0.0f;
            nd.N()[i] = 0u; nd.W()[i] = 0.0f; node_raw<G>(nd)[i] = mask;
            nd.P()[i] = any_win ? mask / (float)nt : 1.0f / (float)nt;
            nd.child()[i] = S.twin[i] ? CHILD_LEAF_WIN : CHILD_LEAF_DRAW; nd.act()[i] = S.tact[i];
        }
        if (tlane<G>() == 0) pn.child()[index] = idx;
        wave_sync();
        backup<G>(E, g, 0, ts, S.path, depth + 1, any_win ? -(float)nt : 0.0f, (uint32_t)nt);
        return false;
    }
    uint8_t h3[3] = {(uint8_t)action, ph.hist3[0], ph.hist3[1]};
    encode_input<G>(S.board, mover, h3, (int)ph.n_hist + 1, E.nn_in + (size_t)g * (G::HW * G::C), E.done_flag != nullptr);
    PathEnt* gp = E.paths + (size_t)g * PathCap<G>::V;
    for (int d = tlane<G>(); d <= depth; d += G::TEAM) gp[d] = S.path[d];
    if (tlane<G>() == 0) {
        gs.pend_kind = PEND_EXPAND; gs.pend_tree = 0; gs.pend_parent = node; gs.pend_slot = index; gs.pend_node = idx;
        gs.pend_depth = depth + 1;
    }
    wave_sync();
    return true;
}

template <class G> GAZ_DEV void g_expand_post(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, Scratch<G>& S,
                                              const float* policy, const float* value_p, bool fresh = false) {
    const int node = gs.pend_parent, index = gs.pend_slot, idx = gs.pend_node, depth = gs.pend_depth;
    NodeRef<G> nd = node_at(E, g, 0, idx);
    g_write_children<G>(E, g, nd, S, policy, fresh);
    NodeRef<G> pn = node_at(E, g, 0, node);
    const float value = *value_p;
    if (tlane<G>() == 0) { pn.child()[index] = idx; node_raw<G>(pn)[index] = value; }       // MCTS_Gumbel.py:516-517
    if (!fresh) {
        const PathEnt* gp = E.paths + (size_t)g * PathCap<G>::V;
        for (int d = tlane<G>(); d < depth; d += G::TEAM) S.path[d] = gp[d];
    }
    wave_sync();
    backup<G>(E, g, 0, ts, S.path, depth, -value, 1u);
}

// end of MCTS_Gumbel.run (MCTS_Gumbel.py:653-679) + Self_Play bookkeeping
template <class G> GAZ_DEV void g_move_end(const DevParams<G>& E, int g, GameState<G>& gs, GumbelState<G>& gu, TreeState& ts, Scratch<G>& S) {
    using RL = RecLayout<G>;
    NodeRef<G> r = g_stage_node<G>(E, g, ts.root, S);                           // LDS copy of the root record
    const int n = tuni<G>((int)r.hdr()->n_actions);
    uint32_t nb; uint64_t sumv;
    compute_pi<G>(E, r, S.raw, n, S, nb, sumv);                                 // pi in S.pri
    uint8_t* rec = rec_of(E, g);
    const int ply = gs.n_hist;
    float* pol = reinterpret_cast<float*>(rec + RL::OFF_POL) + (size_t)ply * G::A;
    uint32_t* rN = reinterpret_cast<uint32_t*>(rec + RL::OFF_N) + (size_t)ply * G::A;
    float* rW = reinterpret_cast<float*>(rec + RL::OFF_W) + (size_t)ply * G::A;
    float* rP = reinterpret_cast<float*>(rec + RL::OFF_P) + (size_t)ply * G::A;
    for (int a = tlane<G>(); a < G::A; a += G::TEAM) { pol[a] = 0.0f; rN[a] = 0u; rW[a] = 0.0f; rP[a] = 0.0f; }
    wave_sync();
    for (int i = tlane<G>(); i < n; i += G::TEAM) {
        const int a = r.act()[i];
        pol[a] = S.pri[i]; rN[a] = r.N()[i]; rW[a] = r.W()[i]; rP[a] = r.P()[i];
    }
    const int top = (gu.n_top > 0) ? (int)gu.top_ids[0] : 0;
    if (tlane<G>() == 0) {
        gs.chosen = r.act()[top];                                               // children[top_node_ids[0]] (MCTS_Gumbel.py:679)
        const uint32_t nv = r.N()[top];
        const float mean = nv > 0 ? (float)((double)r.W()[top] / (double)nv) : S.pri[top];   // unvisited winrate := pi (:666-667)
        reinterpret_cast<float*>(rec + RL::OFF_Q)[ply] = mean;
        reinterpret_cast<uint32_t*>(rec + RL::OFF_RV)[ply] = (uint32_t)ts.root_visits;
        reinterpret_cast<uint32_t*>(rec + RL::OFF_EV)[ply] = gs.move_evals;
    }
    wave_sync();
}

// (fin: see puct_core.hpp game_step_body — the game's epilogue runs inside the phase loop, where the game yields)
template <class G, class Fin> GAZ_DEV void g_game_step_body(const DevParams<G>& E, int g, Scratch<G>& S, GameState<G>& gs, GumbelState<G>& gu, TreeState& ts, Fin&& fin) {
    using RL = RecLayout<G>;

    const long long tp0 = GAZ_PROF_NOW();
    if (E.prof && tlane<G>() == 0) E.prof[(size_t)g * 8 + 7] += 1;
    if (tuni<G>(gs.pend_kind) == PEND_ROOT) {
        g_write_children<G>(E, g, node_at(E, g, 0, ts.root), S, E.nn_policy + (size_t)g * G::A);
        if (tlane<G>() == 0) { gs.pend_kind = PEND_NONE; gs.roots_todo = 0; gs.n_evals += 1; }
        wave_sync();
    } else if (tuni<G>(gs.pend_kind) == PEND_EXPAND) {
        g_expand_post<G>(E, g, gs, ts, S, E.nn_policy + (size_t)g * G::A, E.nn_value + g);
        if (tlane<G>() == 0) {
            gs.pend_kind = PEND_NONE; gs.n_evals += 1; gs.move_evals += 1;
            if (gu.pend_counts) { gu.sims_left -= 1; gu.cur_iter += 1; gs.n_sims += 1; }
        }
        wave_sync();
    }

    GAZ_PROF(0, tp0);
    int tree_only = 0;
    auto phase_step = [&]() -> bool {               // one phase of the state machine; true = the game yields for this launch
        const int phase = tuni<G>(gs.phase);
        if (phase == PH_NEW_GAME) {
            for (int c = tlane<G>(); c < G::BPAD; c += G::TEAM) gs.board[c] = 0;
            if (tlane<G>() == 0) {
                gs.n_hist = 0; gs.next_player = -1; gs.roots_todo = 1; gs.phase = PH_ROOT; gs.winner = RUNNING; gs.host_move = -1;
                gs.move_evals = 0; ts.root = -1; ts.event = 0; ts.n_nodes = 0; ts.root_visits = 0;
            }
            wave_sync();
        } else if (phase == PH_ROOT) {
            if (tuni<G>(gs.roots_todo) == 0) { if (tlane<G>() == 0) gs.phase = PH_MOVE_BEGIN; wave_sync(); return false; }
            if (tlane<G>() == 0) gs.move_evals = 0;
            if (g_root_pre<G>(E, g, gs, ts, S)) {
                const uint8_t* hit = E.cache ? cache_probe<G>(E, g) : nullptr;
                if (hit) {                                                     // evaluation cache hit
                    g_write_children<G>(E, g, node_at(E, g, 0, ts.root), S, reinterpret_cast<const float*>(hit + CacheLayout<G>::OFF_POL), true);
                    if (tlane<G>() == 0) { gs.roots_todo = 0; gs.n_evals += 1; gs.n_hits += 1; }
                    wave_sync();
                    return false;
                }
                if (tlane<G>() == 0) { gs.pend_kind = PEND_ROOT; gs.pend_tree = 0; }
                wave_sync();
                return true;
            }
            if (tlane<G>() == 0) gs.roots_todo = 0;
            wave_sync();
            if (tuni<G>(*E.error)) return true;
        } else if (phase == PH_MOVE_BEGIN) {                                   // head of MCTS_Gumbel.run (:570-599)
            copy_board<G>(S.board, gs.board);
            wave_sync();
            const int len_legal = build_legal<G>(S.board, S.legal);
            NodeRef<G> r = node_at(E, g, 0, ts.root);
            const int n = tuni<G>((int)r.hdr()->n_actions);
            det::Event e = make_event(E, g, gs, ts, 0, det::P_GUMBEL);
            for (int i = tlane<G>(); i < n; i += G::TEAM) {
                // use_gumbel_noise=True (Self_Play.py:64): logits + np.random.gumbel, cast to f32; False (the class default,
                // MCTS_Gumbel.py:157,592-596): the f32 logit priors as they are, and no draw
                gu.top_logits[i] = E.no_gumbel_noise ? r.P()[i] : (float)((double)r.P()[i] + det::gumbel(e, (uint32_t)i));
                gu.top_ids[i] = (uint8_t)i; gu.top_mean[i] = r.W()[i];
            }
            if (tlane<G>() == 0) {
                if (!E.no_gumbel_noise) ts.event += 1;
                gu.m_eff = E.gumbel_m < len_legal ? E.gumbel_m : len_legal; gu.phase = 0; gu.n_top = n; gu.cur_iter = 0;
                gu.iter_limit = E.move_time_ticks ? 3 * len_legal : E.run_iterations;
                gu.cand = 0; gu.stage = 0; gu.sims_left = 0; gu.pend_counts = 0;
            }
            wave_sync();
            bool go = len_legal > 1;
            if (go) go = g_halving<G>(E, gu, r, n, S);
            if (tlane<G>() == 0) gs.phase = go ? PH_SIMS : PH_MOVE_END;
            wave_sync();
        } else if (phase == PH_SIMS) {                                         // MCTS_Gumbel.py:625-648
            NodeRef<G> r = node_at(E, g, 0, ts.root);
            const int n_root = tuni<G>((int)r.hdr()->n_actions);
            if (tuni<G>(gu.cand) >= tuni<G>(gu.n_top)) {                               // phase finished: q-hat of the survivors, halve again
                for (int c = tlane<G>(); c < gu.n_top; c += G::TEAM) {
                    const int id = gu.top_ids[c];
                    gu.top_mean[c] = (float)((double)r.W()[id] / (double)r.N()[id]);
                }
                if (tlane<G>() == 0) gu.phase += 1;
                wave_sync();
                const bool go = g_halving<G>(E, gu, r, n_root, S);
                if (!go) { if (tlane<G>() == 0) gs.phase = PH_MOVE_END; wave_sync(); }
                return false;
            }
            const int id = tuni<G>((int)gu.top_ids[tuni<G>(gu.cand)]);
            if (tuni<G>(gu.stage) == 0) {                                          // expand the root child first (not an iteration)
                if (tlane<G>() == 0) { gu.stage = 1; gu.sims_left = gu.vpc; gu.pend_counts = 0; }
                wave_sync();
                if (tuni<G>(r.child()[id]) == CHILD_NONE) {
                    if (g_expand_pre<G>(E, g, gs, ts, S, ts.root, id, 0)) {
                        const uint8_t* hit = E.cache ? cache_probe<G>(E, g) : nullptr;
                        if (!hit) return true;
                        g_expand_post<G>(E, g, gs, ts, S, reinterpret_cast<const float*>(hit + CacheLayout<G>::OFF_POL),
                                         reinterpret_cast<const float*>(hit + CacheLayout<G>::OFF_VAL), true);   // hit: not an iteration (pend_counts = 0)
                        if (tlane<G>() == 0) { gs.pend_kind = PEND_NONE; gs.n_evals += 1; gs.move_evals += 1; gs.n_hits += 1; }
                        wave_sync();
                    }
                    if (tuni<G>(*E.error)) return true;
                }
                return false;
            }
            if (tuni<G>(gu.sims_left) <= 0) { if (tlane<G>() == 0) { gu.cand += 1; gu.stage = 0; } wave_sync(); return false; }
            if (tree_only >= E.max_tree_sims) return true;
            tree_only++;
            // one simulation below root child `id`
            int node = ts.root, depth = 0, slot = id;
            bool done = false, pending = false, staged = false;
            const long long td0 = GAZ_PROF_NOW();
            for (;;) {
                // below the root the record of `node` is already in LDS (staged for its deterministic_selection)
                const int c = staged ? tuni<G>(NodeRef<G>{reinterpret_cast<uint8_t*>(S.node)}.child()[slot]) : tuni<G>(node_at(E, g, 0, node).child()[slot]);
                if (depth >= PathCap<G>::V - 1) { set_error(E.error, ERR_PATH_OVERFLOW); return true; }
                if (c == CHILD_NONE) {                                         // expand (node, slot)
                    if (tlane<G>() == 0) gu.pend_counts = 1;
                    wave_sync();
                    const long long te0 = GAZ_PROF_NOW();
                    pending = g_expand_pre<G>(E, g, gs, ts, S, node, slot, depth, staged);
                    GAZ_PROF(2, te0);
                    done = !pending;
                    break;
                }
                S.path[depth].node = node; S.path[depth].slot = slot; depth++;
                if (c == CHILD_LEAF_WIN || c == CHILD_LEAF_DRAW) {             // terminal child: value 1 / 0 (:635-637)
                    wave_sync();
                    backup<G>(E, g, 0, ts, S.path, depth, c == CHILD_LEAF_WIN ? 1.0f : 0.0f, 1u);
                    done = true;
                    break;
                }
                node = c;
                NodeRef<G> cn = g_stage_node<G>(E, g, node, S); staged = true;
                slot = g_det_select<G>(E, cn, S.raw, tuni<G>((int)cn.hdr()->n_actions), S);
            }
            GAZ_PROF(1, td0);                                              // descent incl. its expand_pre (slot 2 is counted twice)
            if (pending) {
                const long long tc0 = GAZ_PROF_NOW();
                const uint8_t* hit = E.cache ? cache_probe<G>(E, g) : nullptr;
                GAZ_PROF(3, tc0);
                if (!hit) return true;
                const long long tx0 = GAZ_PROF_NOW();
                g_expand_post<G>(E, g, gs, ts, S, reinterpret_cast<const float*>(hit + CacheLayout<G>::OFF_POL),
                                 reinterpret_cast<const float*>(hit + CacheLayout<G>::OFF_VAL), true);   // hit: the simulation completes in this launch
                GAZ_PROF(4, tx0);
                if (tlane<G>() == 0) { gs.pend_kind = PEND_NONE; gs.n_evals += 1; gs.move_evals += 1; gs.n_hits += 1; }
                wave_sync();
                done = true;
            }
            if (tuni<G>(*E.error)) return true;
            if (done && tlane<G>() == 0) { gu.sims_left -= 1; gu.cur_iter += 1; gs.n_sims += 1; }
            wave_sync();
        } else if (phase == PH_MOVE_END) {
            g_move_end<G>(E, g, gs, gu, ts, S);
            if (tlane<G>() == 0) gs.phase = E.sync_moves ? PH_WAIT_HOST : PH_APPLY;
            wave_sync();
            if (E.sync_moves) return true;
        } else if (phase == PH_APPLY) {                                        // Self_Play.py:142-157, new tree every move (:151-153)
            int action = (tuni<G>(gs.host_move) >= 0) ? tuni<G>(gs.host_move) : tuni<G>(gs.chosen);
            if (tuni<G>(gs.n_hist) == 0 && tuni<G>(gs.host_move) < 0) action = opening_override<G>(E, g, gs, action);
            const int mover = tuni<G>(gs.next_player);
            copy_board<G>(S.board, gs.board);
            wave_sync();
            const int cell = landing_cell<G>(S.board, action);
            const bool win = wins_after<G>(S.board, cell, mover);
            const int empties = G::DRAWS ? count_empty<G>(S.board) : 2;
            uint8_t* rec = rec_of(E, g);
            const int ply = gs.n_hist;
            int winner = win ? mover : ((G::DRAWS && empties == 1) ? 0 : RUNNING);
            wave_sync();
            if (tlane<G>() == 0) {
                gs.board[cell] = (int8_t)mover; gs.hist[ply] = (uint8_t)action; rec[RL::OFF_ACT + ply] = (uint8_t)action;
                gs.n_hist = ply + 1; gs.next_player = -mover; gs.host_move = -1; gs.n_plies += 1;
            }
            wave_sync();
            bool ended = winner != RUNNING;
            if (ply + 1 == E.max_actions) { winner = 0; ended = true; }
            if (!ended && tlane<G>() == 0) { gs.roots_todo = 1; gs.phase = (E.sync_moves && E.single_tree) ? PH_IDLE : PH_ROOT; }
            if (ended && tlane<G>() == 0) {
                gs.winner = winner;
                int32_t* hdr = reinterpret_cast<int32_t*>(rec);
                hdr[0] = ply + 1; hdr[1] = winner; hdr[2] = (int32_t)gs.slot_id; hdr[3] = (int32_t)gs.game_seq;
                atomic_max(&E.stats[0], (unsigned long long)(ply + 1));
                atomic_add(&E.stats[1], (unsigned long long)(ply + 1));
                atomic_add(&E.stats[2], 1ull);
                atomic_add(&E.stats[winner + 4], 1ull);
                gs.phase = PH_RING_WAIT;
            }
            wave_sync();
        } else if (phase == PH_RING_WAIT) {
            if (!ring_push<G>(E, g, gs)) return true;
            if (E.sync_moves) return true;
        } else {
            return true;
        }
        return false;
    };
    bool yielded = false;
    for (int guard = 0; guard < 100000; ++guard) {
        if (!yielded && phase_step()) { fin(); yielded = true; }
        if (!ballot(!yielded)) return;              // wave-uniform exit: every team of this wavefront has yielded
    }
    set_error(E.error, ERR_LOOP_GUARD);
    if (!yielded) fin();
}


// The state machine reads and writes its per-game state (phase, counters, candidate list ...) dozens of times per launch, each
// one a dependent global-memory access.  The launch works on an LDS copy instead: one coalesced load on entry, one store on exit.
template <class G> struct GumbelLocal { GameState<G> gs; GumbelState<G> gu; TreeState ts; };

template <class G> GAZ_DEV void g_game_step(const DevParams<G>& E, int g, Scratch<G>& S, GumbelLocal<G>& L, uint32_t* block_rank = nullptr, int block = 0) {
    GameState<G>* gsG = &E.games[g];
    GumbelState<G>* guG = &reinterpret_cast<GumbelState<G>*>(E.gstate)[g];
    TreeState* tsG = &E.trees[(size_t)g * 2];
    if (eval_was_skipped<G>(E, g)) { publish_done<G>(E, g, block_rank, block); return; }      // see puct_core.hpp game_step
    const long long tw0 = GAZ_PROF_NOW();
    copy_state_words<G>(&L.gs, gsG); copy_state_words<G>(&L.gu, guG); copy_state_words<G>(&L.ts, tsG);
    wave_sync();
    g_game_step_body<G>(E, g, S, L.gs, L.gu, L.ts, [&]() {
        wave_sync();
        copy_state_words<G>(gsG, &L.gs); copy_state_words<G>(guG, &L.gu); copy_state_words<G>(tsG, &L.ts);
        publish_done<G>(E, g, block_rank, block);
        GAZ_PROF(6, tw0);
    });
}

}  // namespace gaz
