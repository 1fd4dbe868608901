// puct_core.hpp — the PUCT self-play wave kernel: select / expand / backup, Dirichlet noise, move
// sampling, re-rooting and the Self_Play per-game state machine.  A game is owned by a TEAM of lanes (wave.hpp): the whole
// wavefront for Gomoku, a 16-lane row — four games per wavefront — for Connect4 / TicTacToe; everything below is written
// against tlane / tballot / team_argmax, so the same source serves both (and the one-lane host emulation of tests/emu).
//
// Reference path replaced (all file:line under /root/reference):
//   MCTS._get_best_PUCT_score_index  MCTS.py:172-191   -> best_puct_slot()       (K1)
//   MCTS._PUCT_select                MCTS.py:193-222   -> puct_select()          (K2)
//   MCTS.get_terminal_actions_fn     MCTS.py:247-294   -> terminal_probe()       (K3)
//   MCTS._expand                     MCTS.py:434-511   -> expand_pre()/expand_post() around the batched evaluator (K4,K8,K9)
//   MCTS._expand_with_terminal_actions MCTS.py:367-428 -> make_terminal_parent() (K5)
//   MCTS._back_propagate             MCTS.py:513-526   -> backup()               (K6)
//   MCTS._apply_dirichlet            MCTS.py:243-245   -> make_priors()          (K7)
//   MCTS.create_expand_root          MCTS.py:296-365   -> root_pre()/root_post() (K12)
//   MCTS.run                         MCTS.py:528-618   -> PH_MOVE_BEGIN/PH_SIMS/PH_MOVE_END (K10)
//   MCTS._set_root / prune_tree      MCTS.py:620-671   -> prune()                (K11)
//   Self_Play.__init__ / play        Self_Play.py:37-57,71-157 -> game_step() phases
//   Client_Server.Parallelized_Session/Server (Client_Server.py:10-217) -> deleted: expand_pre writes the
//   leaf's encoded state straight into row g of the wave's evaluator batch in HBM.
//
// One simulation per game is in flight (the reference has no virtual loss), so a "wave" is: every game
// runs tree-only simulations until it needs an evaluation, writes its leaf into the batch and stops; one
// forward pass evaluates all G leaves; the next launch first consumes the result (expand_post + backup)
// and carries on.  Tree updates need no atomics.
#pragma once
#include "det.hpp"
#include "tree.hpp"

namespace gaz {

// a selection path is at most as long as the plies left in the game: MAXT edges (+ the leaf edge, + 1 for the overflow test)
template <class G> struct PathCap { static constexpr int V = (G::MAXT + 3 + 7) / 8 * 8; };

struct PathEnt { int32_t node; int32_t slot; };

template <class G> struct DevParams {
    // configuration
    int32_t n_games, run_iterations, max_actions, explore_first, explore_second;
    int32_t create_new_root, sync_moves, nodes_per_tree, ring_cap, use_dirichlet, max_tree_sims;
    double c_init, c_base, alpha, eps;
    double c_visit, c_scale;   // Gumbel (MCTS_Gumbel.py:160-161)
    int32_t gumbel_m, node_bytes, compact;
    int32_t stop_search;       // host clock expired (MCTS.run(time_limit), MCTS.py:560-563): finish the move now
    // train_config["MCTS_time_limit"] (Self_Play.py:35,100-112), 0 = none.  PUCT: a move ends when its own wall clock (100 MHz ticks since
    // MOVE_BEGIN) passes the limit or its iterations are used up, whichever comes first (MCTS.py:559-560).  Gumbel: any limit makes every move run
    // 3 x its legal moves iterations instead of run_iterations, as the reference does ("Time limit isn't allowed for gumbel", MCTS_Gumbel.py:576-578)
    uint64_t move_time_ticks;
    int32_t fast_find_win;     // MCTS(fast_find_win=True): keep only the first winning move of a position (MCTS.py:282-283)
    int32_t g_stablemax;       // Gumbel: activation_fn = "stablemax" in deterministic_selection (Self_Play.py:69)
    int32_t single_tree;       // 1: one tree searches for both players (MCTS used on its own, e.g. Connect4/play.py, Game_Tester.py:480-513)
    double tau;                // < 0: Self_Play's exploration schedule (tau 1 / 0); >= 0: tau fixed by the caller (MCTS(tau=...), update_hyperparams)
    int32_t no_gumbel_noise;   // MCTS_Gumbel(use_gumbel_noise=False)
    uint32_t first_game_seq;   // game_seq of every slot's first game (resume: games already in the replay file)
    long long games_budget;    // > 0: slot g plays its k-th game iff k * n_games + g < games_budget, then halts
    int32_t n_opening, opening_actions[8];   // train_config["opening_actions"] (Self_Play.py:130-140)
    double opening_weights[8];
    float one_minus_eps;
    uint32_t key0, key1, slot_offset;
    // state in HBM
    uint8_t* arena;            // [n_games][2][nodes_per_tree][NodeLayout::SIZE]
    TreeState* trees;          // [n_games][2]
    GameState<G>* games;       // [n_games]
    void* gstate;              // [n_games] GumbelState (Gumbel search only)
    PathEnt* paths;            // [n_games][PathCap<G>::V] path of the pending expansion (root -> leaf edge list)
    uint8_t* recs;             // [n_games][RecLayout::SIZE] game in progress
    uint8_t* ring;             // [ring_cap][RecLayout::SIZE] finished games
    uint32_t* ring_head;       // [2]: produced, consumed
    // evaluator batch, row g = game g
    int8_t* nn_in;             // [n_games][HW*C]
    float* nn_policy;          // [n_games][A]
    float* nn_value;           // [n_games]
    // evaluation cache (0 = off): direct-mapped, entry = [tag u32][pad u32][state row, 8-byte padded][policy f32 x A][value f32]
    uint8_t* cache; uint32_t* cache_lock; uint32_t cache_mask, cache_epoch; int32_t cache_stride;
    unsigned long long* stats; // [8]: game_stats[0..5] (Self_Play.py:181-188), [6] waves, [7] spare
    const double* puct_table;  // [PUCT_TABLE_N][2]: sqrt(pv), c_init + ln((pv + c_base + 1) / c_base)
    unsigned long long* prof;  // diagnostic (GAZ_TREE_PROF=1): [n_games][8] shader-clock cycles per phase of the PUCT kernel, else null
    int32_t* error;            // first error code, 0 = none
    // fused tree + trunk launch (resnet.hip k_wave_trunk): a game's team publishes "my leaf row of this wave is in memory" so that the trunk
    // workgroups of the SAME launch can start on their boards while slower games are still searching.  Null = plain launches.
    uint32_t* done_flag;       // [n_games] epoch of the last launch that finished the game's tree step
    uint32_t wave_epoch;
    // bounded hand-over (trunk.hpp TrunkArgs::eval_done): a trunk workgroup that takes its boards on writes the launch's epoch here; one whose
    // wait for done_flag ran out of time leaves them unevaluated and writes nothing.  With eval_done set (= the PREVIOUS wave marked its rows), a
    // game whose mark is not that wave's epoch keeps its request pending (the leaf row is still in nn_in) instead of consuming outputs that were
    // never computed.  A game only loses a wave: its results cannot change.  (Marking success, not failure: a workgroup of THIS launch may give
    // up before this game's team has even started, and must not be able to disturb what the team reads about the previous launch.)
    uint32_t* eval_done;       // [n_games] or null
    // completion queue of the fused launch (round 3; trunk.hpp TrunkArgs::queue): instead of a flag per game, a finished game's team appends the
    // game to a queue and the trunk workgroups take ENTRIES, not fixed boards — the first tiles get whichever games finished first, and no
    // workgroup sits on a slot waiting for one slow game (rows of an evaluator batch are independent bit for bit, so which tile evaluates a
    // game cannot change its outputs).  No global atomics: a tree block ranks its own games by a counter in LDS, and entry (rank r, block j)
    // sits at r * n_full + min(r, rem) + j — rank-major, so the entries of every block's fastest game come first.  Entry = epoch << 32 | game.
    int32_t handoff_release;           // 1: the leaf row was written with plain stores -> agent-scope release before the flag / entry (Gomoku: rows not dword-aligned)
    unsigned long long* done_queue;    // [n_games] or null (then done_flag is used)
    int32_t queue_gpb, queue_nfull, queue_rem;   // games per tree block; blocks with that many games; games of the last, partial block
};

// stores / loads that meet at the device's point of coherence (no L1 / per-XCD L2 copy): used for the leaf rows and the done flags
// a fused launch hands from a tree team to a trunk workgroup on another CU / XCD while both are running
GAZ_DEV void store_coherent(int* p, int v) {
#ifdef GAZ_HOST_EMU
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// phase accounting of the PUCT kernel: 0 consume, 1 select, 2 expand_pre, 3 probe, 4 expand_post (hit), 5 terminal backup, 6 whole launch, 7 launches
#ifdef GAZ_HOST_EMU
#define GAZ_PROF(k, t0) do { } while (0)
#define GAZ_PROF_NOW() 0ll
#else
#define GAZ_PROF_NOW() (E.prof ? (long long)clock64() : 0ll)
#define GAZ_PROF(k, t0) do { if (E.prof && tlane<G>() == 0) E.prof[(size_t)g * 8 + (k)] += (unsigned long long)((long long)clock64() - (t0)); } while (0)
#endif

enum : int32_t { ERR_ARENA_FULL = 1, ERR_ROOT_NOT_EXPANDED = 2, ERR_PATH_OVERFLOW = 3, ERR_LOOP_GUARD = 4, ERR_BAD_SELECT = 5 };

template <class G> struct Scratch {   // per-wave LDS
    int8_t board[G::BPAD];
    uint8_t legal[G::APAD];
    uint8_t tact[G::APAD];     // terminal actions, wins first
    uint8_t twin[G::APAD];     // 1 = win, 0 = draw
    uint8_t sact[G::APAD];     // sorted actions
    float pri[G::APAD];
    float spri[G::APAD];
    float aux[G::APAD];
    double gam[G::APAD];
    PathEnt path[PathCap<G>::V];
    alignas(16) uint32_t node[NodeLayout<G>::SIZE / 4];  // the WHOLE record of the node being scored — header, child arrays and board —
                                                         // in one round trip of 16-byte loads per level (expand_pre takes the board from here)
    float raw[G::APAD];        // Gumbel: the RAW[] block of that node (evaluator values of the expanded children)
};

template <class T> GAZ_DEV T uni(T v) {
#ifdef GAZ_HOST_EMU
    return v;
#else
    return (T)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

template <class G> GAZ_DEV NodeRef<G> node_at(const DevParams<G>& E, int g, int t, int idx) {
    const size_t half = E.compact ? (size_t)E.trees[(size_t)g * 2 + t].half : 0;   // arena = [game][tree][half][node]
    size_t off = ((((size_t)g * 2 + t) * (E.compact ? 2 : 1) + half) * (size_t)E.nodes_per_tree + (size_t)idx) * (size_t)E.node_bytes;
    return NodeRef<G>{E.arena + off};
}

GAZ_DEV void set_error(int32_t* err, int32_t code) {      // any lane of any team may report (same value: a benign race)
    if (*err == 0) *err = code;
}

template <class G> GAZ_DEV det::Event make_event(const DevParams<G>& E, int g, const GameState<G>& gs, TreeState& ts,
                                                 int tree, uint32_t purpose) {
    det::Event e;
    e.key0 = E.key0; e.key1 = E.key1; e.slot = gs.slot_id; e.game_seq = gs.game_seq;
    e.event = ts.event; e.tree = (uint32_t)tree; e.purpose = purpose;
    return e;
}

// ---------------------------------------------------------------------------------------------------
// legal actions of a board in ascending action order -> S.legal; returns count.  Lanes test cells,
// ballot + prefix popcount compacts (Connect4.py:271-276, Gomoku.py:114-119, Tictactoe.py:186-187).
template <class G> GAZ_DEV int build_legal(const int8_t* board, uint8_t* out) {
    int n = 0;
    for (int base = 0; base < G::A; base += G::TEAM) {
        int a = base + tlane<G>();
        bool ok = (a < G::A) && action_legal<G>(board, a);
        uint64_t m = tballot<G>(ok);
        if (ok) out[n + popcll(m & ((1ull << tlane<G>()) - 1ull))] = (uint8_t)a;
        n += popcll(m);
    }
    wave_sync();
    return n;
}

template <class G> GAZ_DEV int count_empty(const int8_t* board) {
    int n = 0;
    for (int base = 0; base < G::HW; base += G::TEAM) {
        int c = base + tlane<G>();
        n += popcll(tballot<G>(c < G::HW && board[c] == 0));
    }
    return n;
}

// fast_find_win (MCTS.py:282-283, MCTS_Gumbel.py:313): the scan stops at the first winning move in legal-action order.  Draws found
// before it would be kept too, but a draw needs the last empty cell, i.e. a single legal move, so the result is that one move.
template <class G> GAZ_DEV bool first_win_only(const int8_t* board, const uint8_t* legal, int n_legal, int player, uint8_t* tact, uint8_t* twin) {
    for (int base = 0; base < n_legal; base += G::TEAM) {
        const int i = base + tlane<G>();
        const bool win = i < n_legal && wins_after<G>(board, landing_cell<G>(board, legal[i]), player);
        const uint64_t m = tballot<G>(win);
        if (m) {
            const int first = base + ffsll0(m);
            if (tlane<G>() == 0) { tact[0] = legal[first]; twin[0] = 1; }
            wave_sync();
            return true;
        }
    }
    return false;
}

// K3: which legal actions end the game for `player`?  One lane per candidate; result list is ordered the
// way the reference orders it: stable ascending argsort of the 0/1 mask, reversed (MCTS.py:293-294 with the
// documented tie rule) = wins in DESCENDING candidate order, then draws in descending candidate order.
template <class G> GAZ_DEV int terminal_probe(const int8_t* board, const uint8_t* legal, int n_legal, int player,
                                              uint8_t* tact, uint8_t* twin, bool& any_win, bool fast_find_win = false) {
    if (fast_find_win && first_win_only<G>(board, legal, n_legal, player, tact, twin)) { any_win = true; return 1; }
    const int empties = G::DRAWS ? count_empty<G>(board) : 0;
    int nt = 0;
    // pass 1: wins (descending), pass 2: draws (descending)
    any_win = false;
    for (int pass = 0; pass < ((G::DRAWS && empties == 1) ? 2 : 1); ++pass) {   // a draw needs the last empty cell (wave-uniform)
        for (int base = ((n_legal - 1) / G::TEAM) * G::TEAM; base >= 0; base -= G::TEAM) {
            int i = base + tlane<G>();
            bool hit = false;
            if (i < n_legal) {
                int a = legal[i];
                bool win = wins_after<G>(board, landing_cell<G>(board, a), player);
                hit = (pass == 0) ? win : (!win && empties == 1);
            }
            uint64_t m = tballot<G>(hit);
            if (hit) {
                int higher = popcll(m >> tlane<G>()) - 1;   // hits in higher lanes come first
                tact[nt + higher] = legal[i];
                twin[nt + higher] = (pass == 0) ? 1 : 0;
            }
            nt += popcll(m);
        }
        if (pass == 0) any_win = nt > 0;
    }
    wave_sync();
    return nt;
}

// get_input_state_MCTS -> int8 [HW][C] row of the evaluator batch.
// Connect4.py:329-346 (incl. the plane-0 overwrite once >= 4 moves were played), Gomoku.py:175-177,
// Tictactoe.py:231-235.  hist3 = last three actions newest first, n_hist = len(action_history).
template <class G> GAZ_DEV void encode_input(const int8_t* board, int current_player, const uint8_t* hist3, int n_hist,
                                             int8_t* out, bool coherent = false) {
    if (G::ID == GAME_C4) {
        int max_length = n_hist - 1; if (max_length > 3) max_length = 3; if (max_length < 0) max_length = 0;
        int rem[3] = {-1, -1, -1};
        for (int i = 0; i < max_length; ++i) {      // y = min(where(prev_board[:, x] != 0)), then clear it
            const int x = hist3[i];
#ifdef GAZ_HOST_EMU
            int y = 0;
            for (; y < 6; ++y) {
                int c = y * 7 + x;
                if (board[c] != 0 && c != rem[0] && c != rem[1] && c != rem[2]) break;
            }
#else
            // lanes 0..5 look at one row each; the topmost occupied, not yet removed cell is the lowest set bit (6 = none)
            const int c = (tlane<G>() < 6 ? tlane<G>() : 0) * 7 + x;
            const uint64_t occ = tballot<G>(tlane<G>() < 6 && board[c] != 0 && c != rem[0] && c != rem[1] && c != rem[2]);
            const int y = occ ? ffsll0(occ) : 6;
#endif
            rem[i] = y * 7 + x;
        }
        for (int c = tlane<G>(); c < G::HW; c += G::TEAM) {
            int8_t b = board[c];
            int8_t p2 = (c == rem[0]) ? 0 : b;
            int8_t p1 = (c == rem[0] || c == rem[1]) ? 0 : b;
            int8_t p0 = (max_length >= 3) ? ((c == rem[0] || c == rem[1] || c == rem[2]) ? (int8_t)0 : b)
                                          : (int8_t)current_player;
            if (max_length < 1) p2 = 0;
            if (max_length < 2) p1 = 0;
            char4 v; v.x = p0; v.y = p1; v.z = p2; v.w = b;
            if (coherent) { int w; memcpy(&w, &v, 4); store_coherent(reinterpret_cast<int*>(out + c * 4), w); }
            else *reinterpret_cast<char4*>(out + c * 4) = v;
        }
    } else {
        for (int c = tlane<G>(); c < G::HW; c += G::TEAM) {
            out[c * 2] = (int8_t)(-current_player);
            out[c * 2 + 1] = board[c];
        }
    }
}

// K1: argmax_i  Q_i + U_i over the node's n_actions children (expanded and not), float64 like numpy:
//   U_i = (P_i * (sqrt(Np) / (N_i + 1))) * (c_init + ln((Np + c_base + 1) / c_base));  Q_i = f32(W_i / N_i) or W_i
// The two factors that depend only on the parent's visit count come from a table filled at engine creation by the same code
// (k_init_puct_table): a float64 log, a division and a square root are ~110 of the ~300 vector instructions of a level, and the
// tree kernel is VALU-bound.
constexpr int PUCT_TABLE_N = 16384;
GAZ_DEV double puct_c_of(double pv, double c_init, double c_base) { return c_init + det::dlog((pv + c_base + 1.0) / c_base); }

// the two factors that depend only on the parent's visit count (table, or computed): loaded by the caller BEFORE it stages the node
// record, so that the table's L2 round trip overlaps the record's
GAZ_DEV void puct_factors(uint64_t parent_visits, double c_init, double c_base, const double* table, double& s, double& c) {
    if (table && parent_visits < (uint64_t)PUCT_TABLE_N) { s = table[2 * parent_visits]; c = table[2 * parent_visits + 1]; }
    else { const double pv = (double)parent_visits; s = dsqrt(pv); c = puct_c_of(pv, c_init, c_base); }
}

template <class G> GAZ_DEV int best_puct_slot(const NodeRef<G>& nd, int n_actions, double s, double c) {
    const uint32_t* N = nd.N(); const float* Wv = nd.W(); const float* P = nd.P();
    double best = 0.0; int bi = 0x7fffffff;
    for (int i = tlane<G>(); i < n_actions; i += G::TEAM) {
        uint32_t n = N[i]; float w = Wv[i];
        double u = ((double)P[i] * (s / (double)(n + 1u))) * c;
        float q = w;
        if (n > 0) q = (float)((double)w / (double)n);
        double sc = (double)q + u;
        if (bi == 0x7fffffff || sc > best) { best = sc; bi = i; }
    }
    team_argmax<G>(best, bi);
    return tuni<G>(bi);
}

// K8 + K7 + K9: gather legal policy entries, renormalise (numpy pairwise float32 sum), mix Dirichlet noise,
// sort descending (ties: higher original index first) into S.sact / S.spri.
template <class G> GAZ_DEV void make_priors(const DevParams<G>& E, int g, const GameState<G>& gs, TreeState& ts, int tree,
                                            Scratch<G>& S, const float* policy, int n_legal) {
    for (int i = tlane<G>(); i < n_legal; i += G::TEAM) S.pri[i] = policy[S.legal[i]];
    wave_sync();
    const float sum = det::np_pairwise_sum<float>(S.pri, n_legal);   // uniform: every lane computes the same value
    wave_sync();
    if (E.use_dirichlet) {
        det::Event e = make_event(E, g, gs, ts, tree, det::P_DIRICHLET);
        for (int i = tlane<G>(); i < n_legal; i += G::TEAM) S.gam[i] = det::gamma(e, (uint32_t)i, E.alpha);
        wave_sync();
        double gs_sum = 0.0;                                             // sequential, same in every lane
        if (n_legal <= 8) {                                              // S.gam has >= 8 elements: batch the reads
            double gv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) gv[i] = S.gam[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i < n_legal) gs_sum = gs_sum + gv[i];
        } else {
            for (int i = 0; i < n_legal; ++i) gs_sum = gs_sum + S.gam[i];
        }
        for (int i = tlane<G>(); i < n_legal; i += G::TEAM) {
            float p = S.pri[i] / sum;
            float a = E.one_minus_eps * p;
            S.pri[i] = (float)((double)a + E.eps * (S.gam[i] / gs_sum));
        }
        if (tlane<G>() == 0) ts.event += 1;
    } else {
        for (int i = tlane<G>(); i < n_legal; i += G::TEAM) S.pri[i] = S.pri[i] / sum;
    }
    wave_sync();
    if (n_legal <= 8) {                                   // rank sort, small node: the eight priors in registers first
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = S.pri[j];
        for (int i = tlane<G>(); i < n_legal; i += G::TEAM) {
            const float v = S.pri[i]; int rank = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) rank += (j < n_legal) && ((pv[j] > v) || (pv[j] == v && j > i));
            S.sact[rank] = S.legal[i]; S.spri[rank] = v;
        }
    } else {
        for (int i = tlane<G>(); i < n_legal; i += G::TEAM) {
            float v = S.pri[i]; int rank = 0;
            for (int j = 0; j < n_legal; ++j) { float o = S.pri[j]; rank += (o > v) || (o == v && j > i); }
            S.sact[rank] = S.legal[i]; S.spri[rank] = v;
        }
    }
    wave_sync();
}

template <class G> GAZ_DEV void write_children_from_scratch(const NodeRef<G>& nd, const Scratch<G>& S, int n) {
    for (int i = tlane<G>(); i < n; i += G::TEAM) {
        nd.N()[i] = 0u; nd.W()[i] = 0.0f; nd.P()[i] = S.spri[i]; nd.child()[i] = CHILD_NONE; nd.act()[i] = S.sact[i];
    }
}

template <class G> GAZ_DEV void copy_board(int8_t* dst, const int8_t* src) {
    for (int c = tlane<G>(); c < G::BPAD; c += G::TEAM) dst[c] = src[c];
}

// K6: add (value, visits) along the recorded path, sign alternating upward from the leaf edge.
template <class G> GAZ_DEV void backup(const DevParams<G>& E, int g, int t, TreeState& ts, const PathEnt* path, int depth,
                                       float value, uint32_t visits) {
    for (int d = tlane<G>(); d < depth; d += G::TEAM) {
        NodeRef<G> nd = node_at(E, g, t, path[d].node);
        float v = ((depth - 1 - d) & 1) ? -value : value;
        int s = path[d].slot;
        nd.W()[s] = nd.W()[s] + v;
        nd.N()[s] = nd.N()[s] + visits;
    }
    if (tlane<G>() == 0) ts.root_visits += visits;
    wave_sync();
}

// terminal parent / terminal root record (K5 and the terminal branch of K12)
template <class G> GAZ_DEV void write_terminal_children(const NodeRef<G>& nd, const Scratch<G>& S, int nt, bool any_win,
                                                        bool as_root) {
    for (int i = tlane<G>(); i < nt; i += G::TEAM) {
        float mask = S.twin[i] ? 1.0f : 0.0f;
        nd.N()[i] = 1u;
        // K5: child_values = terminal_mask (MCTS.py:398); K12: every child backed up with value = any_win (MCTS.py:316-344)
        nd.W()[i] = as_root ? (any_win ? 1.0f : 0.0f) : mask;
        nd.P()[i] = any_win ? mask / (float)nt : 1.0f / (float)nt;     // MCTS.py:376-382 / 318-321
        nd.child()[i] = S.twin[i] ? CHILD_LEAF_WIN : CHILD_LEAF_DRAW;
        nd.act()[i] = S.tact[i];
    }
}

template <class G> GAZ_DEV int alloc_node(const DevParams<G>& E, TreeState& ts) {
    int idx = (int)ts.n_nodes;
    if (idx >= E.nodes_per_tree) { set_error(E.error, ERR_ARENA_FULL); return -1; }
    wave_sync();
    if (tlane<G>() == 0) ts.n_nodes = (uint32_t)idx + 1u;
    wave_sync();
    return idx;
}

// K12 first half: create_expand_root for tree t at the game's current position.  Returns true when the
// root needs an evaluation (input row written), false when it was completed here (terminal root).
template <class G> GAZ_DEV bool root_pre(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, int t, Scratch<G>& S) {
    if (tlane<G>() == 0) { ts.n_nodes = 0; ts.root = -1; ts.root_visits = 0; }
    wave_sync();
    copy_board<G>(S.board, gs.board);
    wave_sync();
    const int n_legal = build_legal<G>(S.board, S.legal);
    bool any_win;
    const int nt = terminal_probe<G>(S.board, S.legal, n_legal, gs.next_player, S.tact, S.twin, any_win, E.fast_find_win != 0);
    uint8_t h3[3];
    for (int i = 0; i < 3; ++i) h3[i] = (gs.n_hist - 1 - i >= 0) ? gs.hist[gs.n_hist - 1 - i] : 0;
    const int idx = alloc_node(E, ts);
    if (idx < 0) return false;
    NodeRef<G> nd = node_at(E, g, t, idx);
    if (tlane<G>() == 0) {
        NodeHdr h; memset(&h, 0, sizeof(h));
        h.parent = -1; h.slot = 0; h.player = (int8_t)(-gs.next_player); h.n_hist = (uint16_t)gs.n_hist;
        h.hist3[0] = h3[0]; h.hist3[1] = h3[1]; h.hist3[2] = h3[2]; h.action = h3[0];
        if (nt > 0) { h.n_actions = (uint8_t)nt; h.n_children = (uint8_t)nt; h.flags = NF_TERMINAL_PARENT; }
        *nd.hdr() = h;
        ts.root = idx;
    }
    copy_board<G>(nd.board(), S.board);
    if (nt > 0) {
        write_terminal_children<G>(nd, S, nt, any_win, true);
        if (tlane<G>() == 0) ts.root_visits = (uint64_t)nt;     // one backup per terminal child (MCTS.py:344)
        wave_sync();
        return false;
    }
    encode_input<G>(S.board, -gs.next_player, h3, gs.n_hist, E.nn_in + (size_t)g * (G::HW * G::C), E.done_flag != nullptr);
    wave_sync();
    return true;
}

// K12 second half: the evaluator's policy row -> root priors (the value is discarded, MCTS.py:346-365)
// `policy`: row g of the evaluator output, or the policy block of an evaluation-cache entry
template <class G> GAZ_DEV void root_post(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, int t, Scratch<G>& S,
                                          const float* policy) {
    NodeRef<G> nd = node_at(E, g, t, ts.root);
    copy_board<G>(S.board, nd.board());
    wave_sync();
    const int n_legal = build_legal<G>(S.board, S.legal);
    make_priors<G>(E, g, gs, ts, t, S, policy, n_legal);
    write_children_from_scratch<G>(nd, S, n_legal);
    if (tlane<G>() == 0) { nd.hdr()->n_actions = (uint8_t)n_legal; nd.hdr()->n_children = 0; }
    wave_sync();
}

// K2: descend from the root.  Returns 0 = expand `node` (its next un-popped child), 1 = terminal leaf hit
// (value in leaf_win).  S.path receives the edge list root..selected, depth its length.
template <class G> GAZ_DEV int puct_select(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, int t, Scratch<G>& S,
                                           int& node, int& depth, bool& leaf_win) {
    node = ts.root; depth = 0;
    uint64_t pv = ts.root_visits;
    for (;;) {
        // Stage the node's header + N/W/P/child/action blocks in LDS with ONE coalesced read of the record, so a level
        // of the descent costs one dependent HBM/L2 round trip instead of three (header -> stats -> chosen child).
        double fs, fc;
        puct_factors(pv, E.c_init, E.c_base, E.puct_table, fs, fc);       // issued first: in flight together with the record
        {
            const uint4* src = reinterpret_cast<const uint4*>(node_at(E, g, t, node).p);
            constexpr int NQ = NodeLayout<G>::SIZE / 16;
            static_assert(NodeLayout<G>::SIZE % 16 == 0, "records are copied in 16-byte units");
            wave_sync();
            for (int i = tlane<G>(); i < NQ; i += G::TEAM) reinterpret_cast<uint4*>(S.node)[i] = src[i];
            wave_sync();
        }
        NodeRef<G> nd{reinterpret_cast<uint8_t*>(S.node)};
        const NodeHdr h = *nd.hdr();
        const int n_actions = tuni<G>((int)h.n_actions), n_children = tuni<G>((int)h.n_children);
        if (depth >= PathCap<G>::V - 1) { set_error(E.error, ERR_PATH_OVERFLOW); return -1; }
        if (tuni<G>((int)h.flags) & NF_TERMINAL_PARENT) {           // MCTS.py:200-208
            // np.sum(child_values) > 0  <=>  some child has W > 0 (all W >= 0 here)
            uint64_t winmask_any = 0; int n_win = 0;
            for (int base = 0; base < n_children; base += G::TEAM) {
                int i = base + tlane<G>();
                bool pos = (i < n_children) && nd.W()[i] > 0.0f;
                winmask_any |= tballot<G>(pos);
                n_win += popcll(tballot<G>((i < n_children) && nd.child()[i] == CHILD_LEAF_WIN));
            }
            const bool wins_only = winmask_any != 0;
            const int n_cand = wins_only ? n_win : n_children;
            det::Event e = make_event(E, g, gs, ts, t, det::P_TERMINAL_PICK);
            const int k = (int)det::pick(e, (uint32_t)n_cand);
            if (tlane<G>() == 0) ts.event += 1;
            // k-th candidate in child order
            int slot = -1, seen = 0;
            for (int base = 0; base < n_children && slot < 0; base += G::TEAM) {
                int i = base + tlane<G>();
                bool c = (i < n_children) && (!wins_only || nd.child()[i] == CHILD_LEAF_WIN);
                uint64_t m = tballot<G>(c);
                int cnt = popcll(m);
                if (k < seen + cnt) {
                    int want = k - seen;       // want-th set bit of m
                    uint64_t mm = m;
                    for (int q = 0; q < want; ++q) mm &= mm - 1;
                    slot = base + ffsll0(mm);
                }
                seen += cnt;
            }
            S.path[depth].node = node; S.path[depth].slot = slot; depth++;
            leaf_win = tuni<G>(nd.child()[slot]) == CHILD_LEAF_WIN;
            wave_sync();
            return 1;
        }
        const int best = best_puct_slot<G>(nd, n_actions, fs, fc);
        if (best == n_children) { wave_sync(); return 0; }          // MCTS.py:217-218
        if (best > n_children) { set_error(E.error, ERR_BAD_SELECT); return -1; }
        S.path[depth].node = node; S.path[depth].slot = best; depth++;
        pv = tuni<G>(nd.N()[best]);
        node = tuni<G>(nd.child()[best]);
    }
}

// ---------------------------------------------------------------------------------------------------
// On-device evaluation cache (replaces Session_Cache.Cache_Wrapper, Session_Cache.py:4-26): encoded leaf state -> the evaluator's
// policy / value for it.  Both trees of a game (and thousands of games) evaluate the same positions: in steady state a
// third of the requests repeat (tools/dup_probe2.py).  A hit is consumed inside the same launch, so it costs no wave.
// Tree kernels only READ the table; entries are written by k_cache_insert between the evaluator pass and the next tree
// launch (kernel boundaries order the two), one writer per slot and wave (epoch lock).  The full state is stored and
// compared, so a hit is exact; rows of an evaluator batch are independent, so the cached bits equal a fresh evaluation.
template <class G> struct CacheLayout {
    static constexpr int ROWB = G::HW * G::C, KEYB = (ROWB + 7) / 8 * 8;
    static constexpr int OFF_KEY = 8, OFF_POL = OFF_KEY + KEYB, OFF_VAL = OFF_POL + 4 * G::A, SIZE = (OFF_VAL + 4 + 63) / 64 * 64;
};

GAZ_DEV uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}

// The row is handled in units of one board cell (C bytes: a dword for Connect4, 16 bits otherwise), one load per unit.
template <int C> struct CellUnitC { typedef uint16_t type; };
template <> struct CellUnitC<4> { typedef uint32_t type; };
template <class G> struct CellUnit { typedef typename CellUnitC<G::C>::type type; };

// order-independent 64-bit hash of the encoded row (each lane mixes the cells it owns, the wave adds)
template <class G> GAZ_DEV uint64_t row_hash(const int8_t* row) {
    typedef typename CellUnit<G>::type U;
    static_assert(sizeof(U) == G::C, "one unit per cell");
    const U* r = reinterpret_cast<const U*>(row);
    uint64_t h = 0;
    for (int c = tlane<G>(); c < G::HW; c += G::TEAM) h += mix64(((uint64_t)(c + 1) << 32) | (uint64_t)r[c]);
    return team_sum_u64<G>(h);
}

// entry holding the outputs for row g of nn_in, or null.  The caller reads policy / value straight from the entry.
template <class G> GAZ_DEV const uint8_t* cache_probe(const DevParams<G>& E, int g) {
    using CL = CacheLayout<G>;
    typedef typename CellUnit<G>::type U;
    const int8_t* row = E.nn_in + (size_t)g * CL::ROWB;
    const uint64_t h = mix64(row_hash<G>(row));
    const uint8_t* ent = E.cache + (size_t)((uint32_t)h & E.cache_mask) * (size_t)E.cache_stride;
    const uint32_t tag = (uint32_t)(h >> 32) | 1u;
    if (tuni<G>(*reinterpret_cast<const uint32_t*>(ent)) != tag) return nullptr;
    const U* r = reinterpret_cast<const U*>(row); const U* k = reinterpret_cast<const U*>(ent + CL::OFF_KEY);
    bool same = true;
    for (int c = tlane<G>(); c < G::HW; c += G::TEAM) same = same && (k[c] == r[c]);
    if (tballot<G>(!same) != 0) return nullptr;
    return ent;
}

// k_cache_insert body: game g's pending request was evaluated this wave -> store (row, outputs)
template <class G> GAZ_DEV void cache_insert(const DevParams<G>& E, int g) {
    using CL = CacheLayout<G>;
    if (tuni<G>(E.games[g].pend_kind) == PEND_NONE) return;
    if (E.eval_done && tuni<G>(E.eval_done[g]) != E.wave_epoch) return;     // this wave's evaluator left the row out: nothing to store
    const int8_t* row = E.nn_in + (size_t)g * CL::ROWB;
    const uint64_t h = mix64(row_hash<G>(row));
    const uint32_t slot = (uint32_t)h & E.cache_mask;
    uint32_t prev = 0;
    if (tlane<G>() == 0) prev = atomic_exch(&E.cache_lock[slot], E.cache_epoch);     // one writer per slot and wave
    prev = tshfl<G>(prev, 0);
    if (prev == E.cache_epoch) return;
    uint8_t* ent = E.cache + (size_t)slot * (size_t)E.cache_stride;
    typedef typename CellUnit<G>::type U;
    const U* r = reinterpret_cast<const U*>(row); U* k = reinterpret_cast<U*>(ent + CL::OFF_KEY);
    for (int c = tlane<G>(); c < G::HW; c += G::TEAM) k[c] = r[c];
    float* pol = reinterpret_cast<float*>(ent + CL::OFF_POL);
    for (int a = tlane<G>(); a < G::A; a += G::TEAM) pol[a] = E.nn_policy[(size_t)g * G::A + a];
    if (tlane<G>() == 0) {
        *reinterpret_cast<float*>(ent + CL::OFF_VAL) = E.nn_value[g];
        *reinterpret_cast<uint32_t*>(ent) = (uint32_t)(h >> 32) | 1u;
    }
}

// K4 first half (+K3, K5): expand the next child of `node`.  Returns true if an evaluation is pending
// (row g written), false if the simulation completed here (terminal parent created and backed up).
// `staged`: puct_select left this node's header + child blocks in S.node (saves two dependent round trips)
template <class G> GAZ_DEV bool expand_pre(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, int t, Scratch<G>& S,
                                           int node, int depth, bool staged) {
    NodeRef<G> pn = node_at(E, g, t, node);
    const NodeRef<G> ps = staged ? NodeRef<G>{reinterpret_cast<uint8_t*>(S.node)} : pn;
    const NodeHdr ph = *ps.hdr();
    const int slot = tuni<G>((int)ph.n_children);
    const int action = tuni<G>((int)ps.act()[slot]);                       // popleft (MCTS.py:437)
    const int mover = -(int)tuni<G>((int)ph.player);
    copy_board<G>(S.board, ps.board());                                 // staged: the record (board included) is already in LDS
    wave_sync();
    const int cell = landing_cell<G>(S.board, action);
    wave_sync();
    if (tlane<G>() == 0) S.board[cell] = (int8_t)mover;                 // do_action_MCTS (MCTS.py:441)
    wave_sync();
    const int n_legal = build_legal<G>(S.board, S.legal);
    bool any_win;
    const int nt = terminal_probe<G>(S.board, S.legal, n_legal, -mover, S.tact, S.twin, any_win, E.fast_find_win != 0);
    const int idx = alloc_node(E, ts);
    if (idx < 0) return false;
    NodeRef<G> nd = node_at(E, g, t, idx);
    if (tlane<G>() == 0) {
        NodeHdr h; memset(&h, 0, sizeof(h));
        h.parent = node; h.slot = (int16_t)slot; h.player = (int8_t)mover; h.n_hist = (uint16_t)(ph.n_hist + 1);
        h.hist3[0] = (uint8_t)action; h.hist3[1] = ph.hist3[0]; h.hist3[2] = ph.hist3[1]; h.action = (uint8_t)action;
        if (nt > 0) { h.n_actions = (uint8_t)nt; h.n_children = (uint8_t)nt; h.flags = NF_TERMINAL_PARENT; }
        *nd.hdr() = h;
    }
    S.path[depth].node = node; S.path[depth].slot = slot;
    wave_sync();
    if (nt > 0) {                                                      // K5, MCTS.py:367-428
        write_terminal_children<G>(nd, S, nt, any_win, false);
        if (tlane<G>() == 0) { pn.child()[slot] = idx; pn.hdr()->n_children = (uint8_t)(slot + 1); }
        wave_sync();
        // value = -(len(terminal_mask)) if any win else 0; visits = len(terminal_mask)  (MCTS.py:373-380, 428)
        backup<G>(E, g, t, ts, S.path, depth + 1, any_win ? -(float)nt : 0.0f, (uint32_t)nt);
        return false;
    }
    copy_board<G>(nd.board(), S.board);
    uint8_t h3[3] = {(uint8_t)action, ph.hist3[0], ph.hist3[1]};
    encode_input<G>(S.board, mover, h3, (int)ph.n_hist + 1, E.nn_in + (size_t)g * (G::HW * G::C), E.done_flag != nullptr);
    // park the path for expand_post
    PathEnt* gp = E.paths + (size_t)g * PathCap<G>::V;
    for (int d = tlane<G>(); d <= depth; d += G::TEAM) gp[d] = S.path[d];
    if (tlane<G>() == 0) {
        gs.pend_kind = PEND_EXPAND; gs.pend_tree = t; gs.pend_parent = node; gs.pend_slot = slot; gs.pend_node = idx;
        gs.pend_depth = depth + 1;
    }
    wave_sync();
    return true;
}

// K4 second half: policy/value row of the leaf -> child record, link into the parent, backup(-value, 1)
// `fresh`: called in the launch that ran expand_pre (evaluation-cache hit): S.board and S.path are still the leaf's
template <class G> GAZ_DEV void expand_post(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, int t, Scratch<G>& S,
                                            const float* policy, const float* value_p, bool fresh) {
    const int node = gs.pend_parent, slot = gs.pend_slot, idx = gs.pend_node, depth = gs.pend_depth;
    NodeRef<G> nd = node_at(E, g, t, idx);
    // Everything this step reads from HBM — the leaf's board, the evaluator's policy row and value, the parked path — is independent:
    // all of it is requested before the first dependent use, one round trip instead of four in a row.
    const float value = *value_p;
    constexpr int PK = (G::A + G::TEAM - 1) / G::TEAM, DK = (PathCap<G>::V + G::TEAM - 1) / G::TEAM;
    float polr[PK]; PathEnt per[DK];
#pragma unroll
    for (int k = 0; k < PK; ++k) { const int a = tlane<G>() + k * G::TEAM; polr[k] = a < G::A ? policy[a] : 0.0f; }
    if (!fresh) {
        const PathEnt* gp = E.paths + (size_t)g * PathCap<G>::V;
#pragma unroll
        for (int k = 0; k < DK; ++k) { const int d = tlane<G>() + k * G::TEAM; if (d < depth) per[k] = gp[d]; }
        copy_board<G>(S.board, nd.board());
    }
#pragma unroll
    for (int k = 0; k < PK; ++k) { const int a = tlane<G>() + k * G::TEAM; if (a < G::A) S.aux[a] = polr[k]; }
    if (!fresh) {
#pragma unroll
        for (int k = 0; k < DK; ++k) { const int d = tlane<G>() + k * G::TEAM; if (d < depth) S.path[d] = per[k]; }
    }
    wave_sync();
    const int n_legal = build_legal<G>(S.board, S.legal);
    make_priors<G>(E, g, gs, ts, t, S, S.aux, n_legal);                 // the policy row, staged in LDS
    write_children_from_scratch<G>(nd, S, n_legal);
    NodeRef<G> pn = node_at(E, g, t, node);
    if (tlane<G>() == 0) {
        nd.hdr()->n_actions = (uint8_t)n_legal; nd.hdr()->n_children = 0;
        pn.child()[slot] = idx; pn.hdr()->n_children = (uint8_t)(slot + 1);
    }
    wave_sync();
    backup<G>(E, g, t, ts, S.path, depth, -value, 1u);                 // MCTS.py:511
}

// Re-root with compaction (the arena is double-buffered): breadth-first copy of the subtree under `root_old` into the other
// half, relabelling child / parent indices; the discarded siblings (MCTS.py:620-655 leaves them to the garbage collector)
// simply stay behind.  Wave-cooperative record copies; returns the new node count or -1.
template <class G> GAZ_DEV int compact_subtree(const DevParams<G>& E, int g, int t, TreeState& ts, int root_old) {
    const size_t oh = ts.half, nh = oh ^ 1, nb = (size_t)E.node_bytes;
    uint8_t* base_o = E.arena + ((((size_t)g * 2 + t) * 2 + oh) * (size_t)E.nodes_per_tree) * nb;
    uint8_t* base_n = E.arena + ((((size_t)g * 2 + t) * 2 + nh) * (size_t)E.nodes_per_tree) * nb;
    const int words = (int)(nb / 16);
    auto copy_rec = [&](int dst, int src) {
        const uint4* s4 = reinterpret_cast<const uint4*>(base_o + (size_t)src * nb);
        uint4* d4 = reinterpret_cast<uint4*>(base_n + (size_t)dst * nb);
        for (int i = tlane<G>(); i < words; i += G::TEAM) d4[i] = s4[i];
    };
    copy_rec(0, root_old);
    wave_sync();
    if (tlane<G>() == 0) reinterpret_cast<NodeHdr*>(base_n)->parent = -1;
    int n_new = 1;
    for (int i = 0; i < n_new; ++i) {
        NodeRef<G> nd{base_n + (size_t)i * nb};
        wave_sync();
        if (tuni<G>((int)nd.hdr()->flags) & NF_TERMINAL_PARENT) continue;          // children are leaf codes, no records
        const int nch = tuni<G>((int)nd.hdr()->n_children);
        for (int s = 0; s < nch; ++s) {
            const int c = tuni<G>(nd.child()[s]);
            if (c < 0) continue;
            if (n_new >= E.nodes_per_tree) { set_error(E.error, ERR_ARENA_FULL); return -1; }
            copy_rec(n_new, c);
            wave_sync();
            if (tlane<G>() == 0) { nd.child()[s] = n_new; reinterpret_cast<NodeHdr*>(base_n + (size_t)n_new * nb)->parent = i; }
            n_new++;
        }
    }
    wave_sync();
    if (tlane<G>() == 0) { ts.half = (uint32_t)nh; ts.root = 0; ts.n_nodes = (uint32_t)n_new; }
    wave_sync();
    return n_new;
}

// K11: prune_tree / _set_root for tree t after `action` was played.  Returns true if the tree must be
// rebuilt with create_expand_root (MCTS.py:661-671).
template <class G> GAZ_DEV bool prune(const DevParams<G>& E, int g, TreeState& ts, int t, int action) {
    if (E.create_new_root || ts.root < 0) return true;
    NodeRef<G> r = node_at(E, g, t, ts.root);
    const int n_children = tuni<G>((int)r.hdr()->n_children);
    int found = -1;
    for (int base = 0; base < n_children; base += G::TEAM) {
        int i = base + tlane<G>();
        uint64_t m = tballot<G>(i < n_children && r.act()[i] == (uint8_t)action);
        if (m) { found = base + ffsll0(m); break; }
    }
    if (found < 0) return true;
    const int c = tuni<G>(r.child()[found]);
    if (c < 0) return true;                                           // (terminal leaf: never pruned on a live game)
    const uint32_t v = tuni<G>(r.N()[found]);
    wave_sync();
    if (E.compact) {
        if (compact_subtree<G>(E, g, t, ts, c) < 0) return false;
        if (tlane<G>() == 0) ts.root_visits = (uint64_t)v;
    } else if (tlane<G>() == 0) { ts.root = c; ts.root_visits = (uint64_t)v; }   // MCTS.py:654-655
    wave_sync();
    return false;
}

template <class G> GAZ_DEV uint8_t* rec_of(const DevParams<G>& E, int g) { return E.recs + (size_t)g * RecLayout<G>::SIZE; }

// end of MCTS.run (MCTS.py:591-613) + the bookkeeping of Self_Play.play (Self_Play.py:114-127)
template <class G> GAZ_DEV void move_end(const DevParams<G>& E, int g, GameState<G>& gs, TreeState& ts, int t, Scratch<G>& S) {
    using RL = RecLayout<G>;
    NodeRef<G> r = node_at(E, g, t, ts.root);
    const int n = tuni<G>((int)r.hdr()->n_children);
    if (n != tuni<G>((int)r.hdr()->n_actions)) { set_error(E.error, ERR_ROOT_NOT_EXPANDED); }
    uint8_t* rec = rec_of(E, g);
    const int ply = gs.n_hist;
    float* pol = reinterpret_cast<float*>(rec + RL::OFF_POL) + (size_t)ply * G::A;
    uint32_t* rN = reinterpret_cast<uint32_t*>(rec + RL::OFF_N) + (size_t)ply * G::A;
    float* rW = reinterpret_cast<float*>(rec + RL::OFF_W) + (size_t)ply * G::A;
    float* rP = reinterpret_cast<float*>(rec + RL::OFF_P) + (size_t)ply * G::A;
    for (int a = tlane<G>(); a < G::A; a += G::TEAM) { pol[a] = 0.0f; rN[a] = 0u; rW[a] = 0.0f; rP[a] = 0.0f; }
    wave_sync();
    unsigned long long sumv = 0;
    for (int i = 0; i < n; ++i) sumv += r.N()[i];
    for (int i = tlane<G>(); i < n; i += G::TEAM) {
        int a = r.act()[i];
        pol[a] = (float)((double)r.N()[i] / (double)sumv);           // prob = N / sum(N)  (MCTS.py:594, Connect4.py:421-424)
        rN[a] = r.N()[i]; rW[a] = r.W()[i]; rP[a] = r.P()[i];
    }
    // K10 move sampling
    int chosen;
    det::Event e = make_event(E, g, gs, ts, t, det::P_MOVE);
    const double u = det::uniform(e);
    if (!gs.tau_on[t]) {                                               // tau == 0: one-hot at first argmax N (MCTS.py:602-604)
        uint32_t bv = 0; int bi = 0x7fffffff;
        for (int i = tlane<G>(); i < n; i += G::TEAM) { uint32_t v = r.N()[i]; if (bi == 0x7fffffff || v > bv) { bv = v; bi = i; } }
        team_argmax_u32<G>(bv, bi);
        chosen = tuni<G>(bi);
    } else {                                                           // tau > 0 (MCTS.py:606-612), float64: N^(1/tau) / visits^(1/tau)
        const double ex = (E.tau > 0.0) ? 1.0 / E.tau : 1.0;           // Self_Play's schedule only ever sets tau = 1
        if (ex == 1.0) {                                               // x ** 1.0 == x exactly
            const double den = (double)ts.root_visits;
            for (int i = tlane<G>(); i < n; i += G::TEAM) S.gam[i] = (double)r.N()[i] / den;
        } else {                                                       // det::dpow stands in for libm pow (< 1e-14 relative: the sampled
            const double den = det::dpow((double)ts.root_visits, ex);  // move differs only if u lands within that of a cdf step)
            for (int i = tlane<G>(); i < n; i += G::TEAM) S.gam[i] = det::dpow((double)r.N()[i], ex) / den;
        }
        wave_sync();
        const double s = det::np_pairwise_sum<double>(S.gam, n);
        double acc = 0.0, last = 0.0;
        for (int i = 0; i < n; ++i) { acc = acc + S.gam[i] / s; last = acc; }
        acc = 0.0; chosen = n - 1;
        for (int i = 0; i < n; ++i) { acc = acc + S.gam[i] / s; if (acc / last > u) { chosen = i; break; } }
        wave_sync();
    }
    if (tlane<G>() == 0) {
        ts.event += 1;
        gs.chosen = r.act()[chosen];
        reinterpret_cast<float*>(rec + RL::OFF_Q)[ply] = (float)((double)r.W()[chosen] / (double)r.N()[chosen]);   // Self_Play.py:117-125
        reinterpret_cast<uint32_t*>(rec + RL::OFF_RV)[ply] = (uint32_t)ts.root_visits;
        reinterpret_cast<uint32_t*>(rec + RL::OFF_EV)[ply] = gs.move_evals;
    }
    wave_sync();
}

// Self_Play.py:130-140: at move 0 the played action is drawn from opening_actions (+ the search's own move with the
// remaining probability) through np.random.choice on the game-level stream (tree 2, event 0).
template <class G> GAZ_DEV int opening_override(const DevParams<G>& E, int g, const GameState<G>& gs, int mcts_action) {
    if (E.n_opening <= 0) return mcts_action;
    int acts[9]; double w[9]; int n = E.n_opening; double sum = 0.0;
    for (int i = 0; i < n; ++i) { acts[i] = E.opening_actions[i]; w[i] = E.opening_weights[i]; sum = sum + w[i]; }
    if (sum < 1.0) { acts[n] = mcts_action; w[n] = 1.0 - sum; n++; }
    det::Event e; e.key0 = E.key0; e.key1 = E.key1; e.slot = gs.slot_id; e.game_seq = gs.game_seq;
    e.event = 0; e.tree = 2; e.purpose = det::P_OPENING;
    const double u = det::uniform(e);
    double cdf[9], acc = 0.0;
    for (int i = 0; i < n; ++i) { acc = acc + w[i]; cdf[i] = acc; }
    for (int i = 0; i < n; ++i) if (cdf[i] / cdf[n - 1] > u) return acts[i];
    return acts[n - 1];
}

// Finished game -> host ring (or dropped when no ring is configured), then restart / halt the slot.  false = ring full.
template <class G> GAZ_DEV bool ring_push(const DevParams<G>& E, int g, GameState<G>& gs) {
    using RL = RecLayout<G>;
    if (E.ring_cap > 0) {
        int slot = -1;
        if (tlane<G>() == 0) {                                          // single consumer (host, between launches) / many producers
            uint32_t prod = atomic_add(&E.ring_head[0], 1u);
            if (prod - E.ring_head[1] < (uint32_t)E.ring_cap) slot = (int)(prod % (uint32_t)E.ring_cap);
            else atomic_add(&E.ring_head[0], (uint32_t)-1);
        }
        slot = tshfl<G>(slot, 0);
        if (slot < 0) return false;                                    // ring full: retry next launch
        const uint4* src = reinterpret_cast<const uint4*>(rec_of(E, g));
        uint4* dst = reinterpret_cast<uint4*>(E.ring + (size_t)slot * RL::SIZE);
        for (int i = tlane<G>(); i < RL::SIZE / 16; i += G::TEAM) dst[i] = src[i];
        wave_sync();
    }
    if (tlane<G>() == 0) {
        if (E.sync_moves) gs.phase = PH_HALT;
        else {
            gs.game_seq += 1;
            // a generation = the first games_budget games STARTED, each run to its end (Self_Play.py:346-408): stop restarting
            const long long k = (long long)(gs.game_seq - E.first_game_seq);
            gs.phase = (E.games_budget > 0 && k * (long long)E.n_games + (long long)(gs.slot_id - E.slot_offset) >= E.games_budget) ? PH_HALT : PH_NEW_GAME;
        }
    }
    wave_sync();
    return true;
}

// One launch of the wave kernel for game g: consume the pending evaluation, then run until the next one.
// `fin` = what the caller does once this game's step is over (state back to HBM, done flag / queue entry of the fused launch).  It runs INSIDE the
// phase loop, at the point where the game yields: with several games per wavefront the code behind a loop only executes once the LAST team has
// left it (the compiler gives a divergent loop one exit), so a game that needed a single simulation used to publish its leaf row when the
// slowest game of its wavefront was done — 53 us instead of 22 at the median (tools/fused_timeline.py).  The loop therefore runs until every
// team of the wavefront has yielded, a team that has yielded sitting out the remaining iterations.
template <class G, class Fin> GAZ_DEV void game_step_body(const DevParams<G>& E, int g, Scratch<G>& S, GameState<G>& gs, TreeState* trees, Fin&& fin) {
    using RL = RecLayout<G>;

    const long long tp0 = GAZ_PROF_NOW();
    if (E.prof && tlane<G>() == 0) E.prof[(size_t)g * 8 + 7] += 1;
    if (tuni<G>(gs.pend_kind) == PEND_ROOT) {
        const int t = tuni<G>(gs.pend_tree);
        root_post<G>(E, g, gs, trees[t], t, S, E.nn_policy + (size_t)g * G::A);
        if (tlane<G>() == 0) { gs.pend_kind = PEND_NONE; gs.roots_todo &= ~(1 << t); gs.n_evals += 1; }
        wave_sync();
    } else if (tuni<G>(gs.pend_kind) == PEND_EXPAND) {
        const int t = tuni<G>(gs.pend_tree);
        expand_post<G>(E, g, gs, trees[t], t, S, E.nn_policy + (size_t)g * G::A, E.nn_value + g, false);
        if (tlane<G>() == 0) { gs.pend_kind = PEND_NONE; gs.sims_done += 1; gs.n_evals += 1; gs.n_sims += 1; gs.move_evals += 1; }
        wave_sync();
    }

    GAZ_PROF(0, tp0);
    int tree_only = 0;   // simulations completed in this launch without an evaluation
    auto phase_step = [&]() -> bool {               // one phase of the state machine; true = the game yields for this launch
        const int phase = tuni<G>(gs.phase);
        if (phase == PH_NEW_GAME) {                                    // Game.__init__ + Self_Play.__init__ (Self_Play.py:37-57)
            for (int c = tlane<G>(); c < G::BPAD; c += G::TEAM) gs.board[c] = 0;
            if (tlane<G>() == 0) {
                gs.n_hist = 0; gs.next_player = -1; gs.roots_todo = E.single_tree ? 1 : 3; gs.phase = PH_ROOT; gs.winner = RUNNING;
                gs.host_move = -1; gs.move_evals = 0;
                trees[0].root = -1; trees[0].event = 0; trees[0].n_nodes = 0; trees[0].root_visits = 0;
                trees[1].root = -1; trees[1].event = 0; trees[1].n_nodes = 0; trees[1].root_visits = 0;
            }
            wave_sync();
        } else if (phase == PH_ROOT) {
            const int todo = tuni<G>(gs.roots_todo);
            if (todo == 0) { if (tlane<G>() == 0) gs.phase = PH_MOVE_BEGIN; wave_sync(); return false; }
            const int t = (todo & 1) ? 0 : 1;
            if (root_pre<G>(E, g, gs, trees[t], t, S)) {
                const uint8_t* hit = E.cache ? cache_probe<G>(E, g) : nullptr;
                if (hit) {                                             // evaluation cache hit: the root is complete in this launch
                    root_post<G>(E, g, gs, trees[t], t, S, reinterpret_cast<const float*>(hit + CacheLayout<G>::OFF_POL));
                    if (tlane<G>() == 0) { gs.roots_todo &= ~(1 << t); gs.n_evals += 1; gs.n_hits += 1; }
                    wave_sync();
                    return false;
                }
                if (tlane<G>() == 0) { gs.pend_kind = PEND_ROOT; gs.pend_tree = t; }
                wave_sync();
                return true;
            }
            if (tlane<G>() == 0) gs.roots_todo &= ~(1 << t);
            wave_sync();
            if (tuni<G>(*E.error)) return true;
        } else if (phase == PH_MOVE_BEGIN) {                           // Self_Play.py:82-106 + MCTS.run head (MCTS.py:542-558)
            copy_board<G>(S.board, gs.board);
            wave_sync();
            const int len_legal = build_legal<G>(S.board, S.legal);
            if (tlane<G>() == 0) {
                const int num = gs.n_hist;
                gs.tau_on[0] = (num % 2 == 0 && num / 2 < E.explore_first) ? 1 : 0;
                gs.tau_on[1] = ((num + 1) % 2 == 0 && (num + 1) / 2 < E.explore_second) ? 1 : 0;
                if (E.tau >= 0.0) { gs.tau_on[0] = E.tau != 0.0; gs.tau_on[1] = gs.tau_on[0]; }
                gs.runner = E.single_tree ? 0 : ((gs.next_player == -1) ? 0 : 1);
                int lim = E.run_iterations;
                if (len_legal == 1) lim = 1; else if (lim < len_legal) lim = len_legal * 3;
                gs.iter_limit = lim; gs.sims_done = 0; gs.fully_visited = 0; gs.move_evals = 0;
                if (E.move_time_ticks) gs.move_t0 = (uint64_t)wall_clock64();
                gs.phase = PH_SIMS;
            }
            wave_sync();
        } else if (phase == PH_SIMS) {                                 // MCTS.run loop body (MCTS.py:560-587)
            // (only once every root child has been visited: stopping earlier, the reference divides by zero visits at MCTS.py:594-595 under its
            // np.seterr(all="raise") — a limit that short is an error there, a floor of one visit per root child here)
            const bool out_of_time = E.move_time_ticks && tuni<G>(gs.fully_visited) && tuni<G>((int)((uint64_t)wall_clock64() - gs.move_t0 > E.move_time_ticks));
            if (tuni<G>(gs.sims_done) >= tuni<G>(gs.iter_limit) || ((E.stop_search || out_of_time) && tuni<G>(gs.sims_done) > 0)) { if (tlane<G>() == 0) gs.phase = PH_MOVE_END; wave_sync(); return false; }
            const int t = tuni<G>(gs.runner);
            TreeState& ts = trees[t];
            NodeRef<G> r = node_at(E, g, t, ts.root);
            if (!tuni<G>(gs.fully_visited)) {                              // 0 not in root.child_visits (MCTS.py:564-565)
                const int na = tuni<G>((int)r.hdr()->n_actions);
                uint64_t zero = 0;
                for (int base = 0; base < na; base += G::TEAM) { int i = base + tlane<G>(); zero |= tballot<G>(i < na && r.N()[i] == 0u); }
                if (!zero) { if (tlane<G>() == 0) gs.fully_visited = 1; wave_sync(); }
            }
            if (tuni<G>(gs.fully_visited) && (tuni<G>((int)r.hdr()->flags) & NF_TERMINAL_PARENT)) {
                // Root with a terminal move available (MCTS.py:200-208 at depth 0): every remaining simulation is
                // "pick a terminal child, back up 1 (win) or 0 (draw)" and touches only the root's arrays, so up to 64
                // of them run at once, one RNG event per lane.  f32 adds of 1.0 onto integral W are exact, so
                // W += count equals the reference's sequence of += 1.
                const int nch = tuni<G>((int)r.hdr()->n_children);
                int n_cand = 0, n_win = 0;
                for (int base = 0; base < nch; base += G::TEAM) {
                    int i = base + tlane<G>();
                    uint64_t m = tballot<G>(i < nch && r.child()[i] == CHILD_LEAF_WIN);
                    if (i < nch && r.child()[i] == CHILD_LEAF_WIN) S.sact[n_win + popcll(m & ((1ull << tlane<G>()) - 1ull))] = (uint8_t)i;
                    n_win += popcll(m);
                }
                uint64_t anypos = 0;
                for (int base = 0; base < nch; base += G::TEAM) { int i = base + tlane<G>(); anypos |= tballot<G>(i < nch && r.W()[i] > 0.0f); }
                const bool wins_only = anypos != 0;
                if (wins_only) n_cand = n_win;
                else { for (int i = tlane<G>(); i < nch; i += G::TEAM) S.sact[i] = (uint8_t)i; n_cand = nch; }
                wave_sync();
                const int remaining = tuni<G>(gs.iter_limit) - tuni<G>(gs.sims_done);
                const int chunk = remaining < G::TEAM ? remaining : G::TEAM;
                int my_slot = -1;
                if (tlane<G>() < chunk) {
                    det::Event e = make_event(E, g, gs, ts, t, det::P_TERMINAL_PICK);
                    e.event += (uint32_t)tlane<G>();
                    my_slot = S.sact[det::pick(e, (uint32_t)n_cand)];
                }
                for (int c = 0; c < n_cand; ++c) {
                    const int slot = S.sact[c];
                    const int cnt = popcll(tballot<G>(my_slot == slot));
                    if (tlane<G>() == 0 && cnt) {
                        if (r.child()[slot] == CHILD_LEAF_WIN) r.W()[slot] = r.W()[slot] + (float)cnt;
                        r.N()[slot] = r.N()[slot] + (uint32_t)cnt;
                    }
                }
                if (tlane<G>() == 0) {
                    ts.root_visits += (uint64_t)chunk; ts.event += (uint32_t)chunk;
                    gs.sims_done += chunk; gs.n_sims += (uint64_t)chunk;
                }
                wave_sync();
                return false;
            }
            if (tree_only >= E.max_tree_sims) return true;                  // bound the launch's tail; resume next wave
            tree_only++;
            int node, depth; bool leaf_win = false; int kind;
            const long long ts0 = GAZ_PROF_NOW();
            if (!tuni<G>(gs.fully_visited)) { node = ts.root; depth = 0; kind = 0; }
            else kind = puct_select<G>(E, g, gs, ts, t, S, node, depth, leaf_win);
            GAZ_PROF(1, ts0);
            if (kind < 0) return true;
            if (kind == 1) {                                           // terminal leaf: value 1 / 0, visits 1 (MCTS.py:573-575)
                const long long tb0 = GAZ_PROF_NOW();
                backup<G>(E, g, t, ts, S.path, depth, leaf_win ? 1.0f : 0.0f, 1u);
                if (tlane<G>() == 0) { gs.sims_done += 1; gs.n_sims += 1; }
                wave_sync();
                GAZ_PROF(5, tb0);
            } else {
                const long long te0 = GAZ_PROF_NOW();
                const bool pending = expand_pre<G>(E, g, gs, ts, t, S, node, depth, kind == 0 && tuni<G>(gs.fully_visited) != 0);
                GAZ_PROF(2, te0);
                if (pending) {
                    const long long tc0 = GAZ_PROF_NOW();
                    const uint8_t* hit = E.cache ? cache_probe<G>(E, g) : nullptr;
                    GAZ_PROF(3, tc0);
                    if (!hit) return true;                                  // miss: the evaluator answers in the next launch
                    const long long tx0 = GAZ_PROF_NOW();
                    expand_post<G>(E, g, gs, ts, t, S, reinterpret_cast<const float*>(hit + CacheLayout<G>::OFF_POL),
                                   reinterpret_cast<const float*>(hit + CacheLayout<G>::OFF_VAL), true);   // hit: consume the cached outputs now
                    if (tlane<G>() == 0) { gs.pend_kind = PEND_NONE; gs.n_evals += 1; gs.move_evals += 1; gs.n_hits += 1; }
                    wave_sync();
                    GAZ_PROF(4, tx0);
                }
                if (tuni<G>(*E.error)) return true;
                if (tlane<G>() == 0) { gs.sims_done += 1; gs.n_sims += 1; }
                wave_sync();
            }
        } else if (phase == PH_MOVE_END) {
            const int t = tuni<G>(gs.runner);
            move_end<G>(E, g, gs, trees[t], t, S);
            if (tlane<G>() == 0) gs.phase = E.sync_moves ? PH_WAIT_HOST : PH_APPLY;
            wave_sync();
            if (E.sync_moves) return true;
        } else if (phase == PH_APPLY) {                                // Self_Play.py:142-157
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
                gs.board[cell] = (int8_t)mover;
                gs.hist[ply] = (uint8_t)action;
                rec[RL::OFF_ACT + ply] = (uint8_t)action;
                gs.n_hist = ply + 1; gs.next_player = -mover; gs.host_move = -1; gs.n_plies += 1;
            }
            wave_sync();
            bool ended = winner != RUNNING;
            int todo = 0;
            if (!ended) {
                if (prune<G>(E, g, trees[0], 0, action)) todo |= 1;   // both trees prune (Self_Play.py:149-150)
                if (!E.single_tree && prune<G>(E, g, trees[1], 1, action)) todo |= 2;
            }
            // Self_Play.py:155-157: reaching max_actions forces winner = 0 — even when that last action won
            if (ply + 1 == E.max_actions) { winner = 0; ended = true; }
            if (!ended && tlane<G>() == 0) { gs.roots_todo = todo; gs.phase = (E.sync_moves && E.single_tree) ? PH_IDLE : PH_ROOT; }
            if (ended) {
                if (tlane<G>() == 0) {
                    gs.winner = winner;
                    int32_t* hdr = reinterpret_cast<int32_t*>(rec);
                    hdr[0] = ply + 1; hdr[1] = winner; hdr[2] = (int32_t)gs.slot_id; hdr[3] = (int32_t)gs.game_seq;
                    atomic_max(&E.stats[0], (unsigned long long)(ply + 1));            // game_stats (Self_Play.py:181-188)
                    atomic_add(&E.stats[1], (unsigned long long)(ply + 1));
                    atomic_add(&E.stats[2], 1ull);
                    atomic_add(&E.stats[winner + 4], 1ull);
                    gs.phase = PH_RING_WAIT;
                }
            }
            wave_sync();
        } else if (phase == PH_RING_WAIT) {                            // hand the finished game to the host ring
            if (!ring_push<G>(E, g, gs)) return true;
            if (E.sync_moves) return true;
        } else {                                                       // PH_WAIT_HOST, PH_HALT
            return true;
        }
        return false;
    };
    bool done = false;
    for (int guard = 0; guard < 100000; ++guard) {
        if (!done && phase_step()) { fin(); done = true; }
        if (!ballot(!done)) return;                 // wave-uniform exit: every team of this wavefront has yielded
    }
    set_error(E.error, ERR_LOOP_GUARD);
    if (!done) fin();
}


// The state machine touches its per-game state (phase, counters, pending request, both tree heads) dozens of times per launch,
// each one a dependent global-memory access.  Without re-root compaction (where node_at() reads TreeState::half from HBM) the
// launch works on an LDS copy: one coalesced load on entry, one store on exit.
template <class G> struct PuctLocal { GameState<G> gs; TreeState ts[2]; };

template <class G, class T> GAZ_DEV void copy_state_words(T* dst, const T* src) {
    static_assert(sizeof(T) % 4 == 0, "word copy");
    uint32_t* d = reinterpret_cast<uint32_t*>(dst); const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    for (int i = tlane<G>(); i < (int)(sizeof(T) / 4); i += G::TEAM) d[i] = s[i];
}

// fused launch: the team's leaf row (coherent stores, encode_input) has reached memory -> publish the epoch
template <class G> GAZ_DEV void publish_done(const DevParams<G>& E, int g, uint32_t* block_rank = nullptr, int block = 0) {
#ifndef GAZ_HOST_EMU
    if (E.done_queue && block_rank) {               // completion queue: this game is the rank-th of its tree block to finish
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (E.handoff_release) {                    // plain row stores: write back this XCD's L2 before the entry can be seen (Guideline 16 R1)
            if (tlane<G>() == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tlane<G>() == 0) {
            const int rank = (int)atomic_add(block_rank, 1u);              // LDS
            const int entry = rank * E.queue_nfull + (rank < E.queue_rem ? rank : E.queue_rem) + block;
            __hip_atomic_store(E.done_queue + entry, ((unsigned long long)E.wave_epoch << 32) | (unsigned)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (!E.done_flag) return;
    if (E.handoff_release) {                        // (see above)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tlane<G>() == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    // every store of this wave — the leaf row included — has been acknowledged before the flag is written.  Inline asm with a memory clobber: the
    // compiler may neither sink a row store below it nor hoist the flag store above it (MI355X_MICROARCH.md, Valid forms: write-through payload ->
    // asm vmcnt(0) -> flag), which a relaxed atomic + the waitcnt builtin alone would not forbid.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tlane<G>() == 0) store_coherent(reinterpret_cast<int*>(E.done_flag + g), (int)E.wave_epoch);
#endif
}

// did the evaluator leave game g's row out in the previous wave (see DevParams::eval_done)?  Team-uniform.
template <class G> GAZ_DEV bool eval_was_skipped(const DevParams<G>& E, int g) {
    if (!E.eval_done) return false;
    return tuni<G>(E.eval_done[g]) != E.wave_epoch - 1;
}

template <class G> GAZ_DEV void game_step(const DevParams<G>& E, int g, Scratch<G>& S, PuctLocal<G>& L, uint32_t* block_rank = nullptr, int block = 0) {
    GameState<G>* gsG = &E.games[g];
    TreeState* tsG = E.trees + (size_t)g * 2;
    if (eval_was_skipped<G>(E, g)) { publish_done<G>(E, g, block_rank, block); return; }      // the request stays pending: same leaf row, evaluated by this wave
    if (E.compact) { game_step_body<G>(E, g, S, *gsG, tsG, [&]() { publish_done<G>(E, g, block_rank, block); }); return; }
    const long long tw0 = GAZ_PROF_NOW();
    copy_state_words<G>(&L.gs, gsG); copy_state_words<G>(&L.ts[0], &tsG[0]); copy_state_words<G>(&L.ts[1], &tsG[1]);
    wave_sync();
    game_step_body<G>(E, g, S, L.gs, L.ts, [&]() {
        wave_sync();
        copy_state_words<G>(gsG, &L.gs); copy_state_words<G>(&tsG[0], &L.ts[0]); copy_state_words<G>(&tsG[1], &L.ts[1]);
        publish_done<G>(E, g, block_rank, block);
        GAZ_PROF(6, tw0);
    });
}

}  // namespace gaz
