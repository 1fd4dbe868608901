// engine.hip — host side of libgaz_engine.so: owns the HBM state, launches the wave kernel + evaluator
// once per simulation wave on one HIP stream, and implements the C ABI of include/gaz_engine.h.
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/gaz_engine.h"
#include "rt.hpp"
#include "puct_core.hpp"
#include "gumbel_core.hpp"
#include "evaluator.hpp"

using namespace gaz;

static std::string g_create_error;

#define HIP_OK(expr)                                                                  \
    do { hipError_t _e = (expr); if (_e != hipSuccess) { return fail(std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)

// ------------------------------------------------------------------------------------------ kernels
// one TEAM of lanes per game of [g0, g1): a whole wavefront (Gomoku), or a 16-lane row — four games per wavefront (wave.hpp "Teams")
template <class G> GAZ_DEV void wave_body(const DevParams<G>& E, int g0, int g1) {
    constexpr int PER = WAVE / G::TEAM;
    GAZ_SHARED Scratch<G> S[PER];
    GAZ_SHARED PuctLocal<G> L[PER];
    const int t = team_in_wave<G>();
    const int g = g0 + block_id() * PER + t;
    if (g < g1) game_step<G>(E, g, S[t], L[t]);
}
template <class G> GAZ_KERNEL k_wave(DevParams<G> E, int g0, int g1) { wave_body<G>(E, g0, g1); }
template <class G> GAZ_KERNEL_TEAMS k_wave_teams(DevParams<G> E, int g0, int g1) { wave_body<G>(E, g0, g1); }

template <class G> GAZ_DEV void wave_body_gumbel(const DevParams<G>& E, int g0, int g1) {
    constexpr int PER = WAVE / G::TEAM;
    GAZ_SHARED Scratch<G> S[PER];
    GAZ_SHARED GumbelLocal<G> L[PER];
    const int t = team_in_wave<G>();
    const int g = g0 + block_id() * PER + t;
    if (g < g1) g_game_step<G>(E, g, S[t], L[t]);
}
template <class G> GAZ_KERNEL k_wave_gumbel(DevParams<G> E, int g0, int g1) { wave_body_gumbel<G>(E, g0, g1); }
template <class G> GAZ_KERNEL_TEAMS k_wave_gumbel_teams(DevParams<G> E, int g0, int g1) { wave_body_gumbel<G>(E, g0, g1); }

// evaluation cache: store (state row, outputs) of every request the evaluator just answered; runs between the evaluator
// pass and the next tree launch, so the tree kernels only ever read the table
template <class G> GAZ_KERNEL k_cache_insert(DevParams<G> E, int g0, int g1) {
    const int g = g0 + block_id();
    if (g < g1) cache_insert<G>(E, g);
}

// sqrt(pv) and the exploration factor of every parent visit count below PUCT_TABLE_N, computed by the code the kernel would run
template <int UNUSED> GAZ_KERNEL_WIDE k_init_puct_table(double* table, double c_init, double c_base, int n) {
#ifdef GAZ_HOST_EMU
    for (int i = 0; i < n; ++i) { table[2 * i] = dsqrt((double)i); table[2 * i + 1] = puct_c_of((double)i, c_init, c_base); }
#else
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { table[2 * i] = dsqrt((double)i); table[2 * i + 1] = puct_c_of((double)i, c_init, c_base); }
#endif
}

template <class G> GAZ_KERNEL k_init_games(DevParams<G> E, int first_seq) {
    const int g = block_id();
    if (g >= E.n_games || lane_id() != 0) return;
    GameState<G>& gs = E.games[g];
    memset(&gs, 0, sizeof(gs));
    gs.phase = (E.games_budget > 0 && (long long)g >= E.games_budget) ? PH_HALT : PH_NEW_GAME;     // fewer games wanted than slots
    gs.pend_kind = PEND_NONE; gs.game_seq = (uint32_t)first_seq; gs.host_move = -1; gs.winner = RUNNING;
    gs.slot_id = E.slot_offset + (uint32_t)g;
    E.trees[g * 2].root = -1; E.trees[g * 2 + 1].root = -1;
}

template <class G> GAZ_KERNEL k_reset_games(DevParams<G> E, const int32_t* slots, int n) {
    const int i = block_id();
    if (i >= n || lane_id() != 0) return;
    const int g = slots ? slots[i] : i;
    if (g < 0 || g >= E.n_games) return;
    GameState<G>& gs = E.games[g];
    const uint32_t seq = gs.game_seq + ((gs.phase == PH_NEW_GAME && gs.n_evals == 0) ? 0u : 1u);
    const uint64_t ne = gs.n_evals, ns = gs.n_sims, np = gs.n_plies; const uint32_t sid = gs.slot_id;
    memset(&gs, 0, sizeof(gs));
    gs.phase = PH_NEW_GAME; gs.game_seq = seq; gs.host_move = -1; gs.winner = RUNNING; gs.n_evals = ne; gs.n_sims = ns; gs.n_plies = np; gs.slot_id = sid;
}

// put one slot at an arbitrary position (MCTS attaches to a live game object, MCTS.py:296-313): replay `n` actions on an empty
// board, then let the state machine build the root(s) there.
template <class G> GAZ_KERNEL k_set_position(DevParams<G> E, int g, const int32_t* actions, int n) {
    if (block_id() != 0 || lane_id() != 0 || g < 0 || g >= E.n_games) return;
    GameState<G>& gs = E.games[g];
    const uint32_t seq = gs.game_seq, sid = gs.slot_id; const uint64_t ne = gs.n_evals, ns = gs.n_sims, np = gs.n_plies;
    memset(&gs, 0, sizeof(gs));
    gs.game_seq = seq; gs.n_evals = ne; gs.n_sims = ns; gs.n_plies = np; gs.host_move = -1; gs.winner = RUNNING; gs.slot_id = sid;
    int player = -1;
    for (int i = 0; i < n && i < G::MAXT; ++i) {
        const int a = actions[i];
        gs.board[landing_cell<G>(gs.board, a)] = (int8_t)player;
        gs.hist[i] = (uint8_t)a; player = -player;
        E.recs[(size_t)g * RecLayout<G>::SIZE + RecLayout<G>::OFF_ACT + i] = (uint8_t)a;
    }
    gs.n_hist = n < G::MAXT ? n : G::MAXT; gs.next_player = player;
    gs.roots_todo = (E.single_tree || E.gstate) ? 1 : 3; gs.phase = (E.sync_moves && E.single_tree) ? PH_IDLE : PH_ROOT; gs.pend_kind = PEND_NONE;
    for (int t = 0; t < 2; ++t) { TreeState& ts = E.trees[(size_t)g * 2 + t]; ts.root = -1; ts.n_nodes = 0; ts.root_visits = 0; ts.event = 0; }
}

// gaz_engine_read_positions: the action history of every slot's game in progress (what gaz_engine_set_position takes)
template <class G> GAZ_KERNEL k_read_positions(DevParams<G> E, int32_t* n_hist, uint8_t* hist, int stride) {
    const int g = block_id();
    if (g >= E.n_games) return;
    const GameState<G>& gs = E.games[g];
    const int n = gs.phase == PH_HALT ? 0 : gs.n_hist;
    for (int i = lane_id(); i < stride; i += WAVE) hist[(size_t)g * stride + i] = i < n ? gs.hist[i] : (uint8_t)0;
    if (lane_id() == 0) n_hist[g] = n;
}

template <class G> GAZ_KERNEL k_start_search(DevParams<G> E) {      // PH_IDLE -> build the missing root(s), then MCTS.run
    const int g = block_id();
    if (g >= E.n_games || lane_id() != 0) return;
    if (E.games[g].phase == PH_IDLE) E.games[g].phase = PH_ROOT;
}

// Game-rules probe (gaz_engine_probe_rules): one wavefront per position, the position given as an action history from the empty
// board.  Runs the DEVICE rule code the search uses — landing_cell / wins_after (do_action_MCTS + check_win_MCTS), build_legal
// (get_legal_actions_MCTS), encode_input (get_input_state_MCTS), terminal_probe (get_terminal_actions_fn), make_priors without
// noise (get_legal_actions_policy_MCTS, normalize=True) — so the reference's Game classes can be compared with it directly.
template <class G> GAZ_KERNEL k_probe_rules(DevParams<G> E, const int32_t* actions, const int32_t* n_actions, int n_pos, int stride,
                                            int8_t* o_board, uint8_t* o_legal, int32_t* o_winner, int8_t* o_input, int32_t* o_term,
                                            const float* policy_in, float* o_policy) {
    GAZ_SHARED Scratch<G> S;
    const int p = block_id();
    if (p >= n_pos) return;
    const int n = n_actions[p];
    for (int c = lane_id(); c < G::BPAD; c += WAVE) S.board[c] = 0;
    wave_sync();
    int player = -1, winner = RUNNING;
    for (int i = 0; i < n; ++i) {
        const int a = actions[(size_t)p * stride + i];
        if (!action_legal<G>(S.board, a)) { if (lane_id() == 0) o_winner[p] = -99; return; }   // not a legal history
        const int cell = landing_cell<G>(S.board, a);
        const bool win = wins_after<G>(S.board, cell, player);
        const int empties = G::DRAWS ? count_empty<G>(S.board) : 2;
        winner = win ? player : ((G::DRAWS && empties == 1) ? 0 : RUNNING);        // check_win after this move
        wave_sync();
        if (lane_id() == 0) S.board[cell] = (int8_t)player;
        wave_sync();
        player = -player;
    }
    for (int c = lane_id(); c < G::HW; c += WAVE) o_board[(size_t)p * G::HW + c] = S.board[c];
    const int n_legal = build_legal<G>(S.board, S.legal);
    for (int a = lane_id(); a < G::A; a += WAVE) { o_legal[(size_t)p * G::A + a] = 0; o_term[(size_t)p * G::A + a] = -1; if (o_policy) o_policy[(size_t)p * G::A + a] = 0.0f; }
    wave_sync();
    for (int i = lane_id(); i < n_legal; i += WAVE) o_legal[(size_t)p * G::A + S.legal[i]] = 1;
    uint8_t h3[3];
    for (int i = 0; i < 3; ++i) h3[i] = (n - 1 - i >= 0) ? (uint8_t)actions[(size_t)p * stride + n - 1 - i] : 0;
    encode_input<G>(S.board, -player, h3, n, o_input + (size_t)p * (G::HW * G::C));
    if (lane_id() == 0) o_winner[p] = winner;
    if (policy_in && o_policy && n_legal > 0) {
        DevParams<G> E2 = E; E2.use_dirichlet = 0;
        make_priors<G>(E2, 0, E.games[0], E.trees[0], 0, S, policy_in + (size_t)p * G::A, n_legal);   // game / tree state: untouched without noise
        for (int i = lane_id(); i < n_legal; i += WAVE) o_policy[(size_t)p * G::A + S.legal[i]] = S.pri[i];
        wave_sync();
    }
    if (winner != RUNNING) return;                                                 // the search never expands a finished position
    bool any_win;
    const int nt = terminal_probe<G>(S.board, S.legal, n_legal, player, S.tact, S.twin, any_win, E.fast_find_win != 0);
    for (int i = lane_id(); i < nt; i += WAVE) o_term[(size_t)p * G::A + S.tact[i]] = S.twin[i] ? 1 : 0;
}

// gaz_engine_repack: move the running game of physical slot `src` into the halted slot `dst` (dst < src).  Everything a game owns
// travels: its tree arena slice, both tree heads, the path buffer, the record in progress, the Gumbel state, the rows of the
// evaluator batch (a pending request's answer is waiting there) — and the GameState is SWAPPED, so the halted slot's lifetime
// counters stay in the sum gaz_engine_get_stats reports.  The game keeps its identity (GameState::slot_id).
template <class G> GAZ_KERNEL_WIDE k_move_slots(DevParams<G> E, const int32_t* moves, int n_moves, size_t arena_slot_bytes, size_t gstate_bytes) {
#ifdef GAZ_HOST_EMU
    const int b = block_id(), T = 1, t = 0;
#else
    const int b = blockIdx.x, T = blockDim.x, t = threadIdx.x;
#endif
    if (b >= n_moves) return;
    const int src = moves[2 * b], dst = moves[2 * b + 1];
    auto copy16 = [&](void* d, const void* s_, size_t bytes) {        // bytes % 16 == 0
        uint4* dd = reinterpret_cast<uint4*>(d); const uint4* ss = reinterpret_cast<const uint4*>(s_);
        for (size_t i = t; i < bytes / 16; i += T) dd[i] = ss[i];
    };
    auto copy1 = [&](void* d, const void* s_, size_t bytes) {
        uint8_t* dd = reinterpret_cast<uint8_t*>(d); const uint8_t* ss = reinterpret_cast<const uint8_t*>(s_);
        for (size_t i = t; i < bytes; i += T) dd[i] = ss[i];
    };
    copy16(E.arena + (size_t)dst * arena_slot_bytes, E.arena + (size_t)src * arena_slot_bytes, arena_slot_bytes);
    copy16(E.recs + (size_t)dst * RecLayout<G>::SIZE, E.recs + (size_t)src * RecLayout<G>::SIZE, RecLayout<G>::SIZE);
    copy1(E.paths + (size_t)dst * PathCap<G>::V, E.paths + (size_t)src * PathCap<G>::V, sizeof(PathEnt) * PathCap<G>::V);
    copy1(&E.trees[(size_t)dst * 2], &E.trees[(size_t)src * 2], 2 * sizeof(TreeState));
    if (E.gstate) copy1(reinterpret_cast<uint8_t*>(E.gstate) + (size_t)dst * gstate_bytes, reinterpret_cast<uint8_t*>(E.gstate) + (size_t)src * gstate_bytes, gstate_bytes);
    copy1(E.nn_in + (size_t)dst * (G::HW * G::C), E.nn_in + (size_t)src * (G::HW * G::C), G::HW * G::C);
    copy1(E.nn_policy + (size_t)dst * G::A, E.nn_policy + (size_t)src * G::A, 4 * G::A);
    copy1(E.nn_value + dst, E.nn_value + src, 4);
    if (E.eval_done) copy1(E.eval_done + dst, E.eval_done + src, 4);
    // swap the two GameStates word by word (each thread its own words: no staging needed)
    uint32_t* ga = reinterpret_cast<uint32_t*>(&E.games[src]); uint32_t* gb = reinterpret_cast<uint32_t*>(&E.games[dst]);
    for (size_t i = t; i < sizeof(GameState<G>) / 4; i += T) { const uint32_t x = ga[i]; ga[i] = gb[i]; gb[i] = x; }
}

// gaz_engine_debug_fused_fault on a path without the fused launch (other games / searches / the CPU emulation build): what a trunk workgroup
// that gave up leaves behind — its boards marked with the wave's epoch, their outputs NOT those of this wave's requests (poisoned here, so that
// a tree step that consumed them could not go unnoticed), the fault counter raised.
template <class G> GAZ_KERNEL k_debug_skip(DevParams<G> E, uint32_t* eval_done, int n, unsigned mod, int32_t* fault) {
    const int g = block_id();
    if (g >= n || lane_id() != 0) return;
    if ((unsigned)(g / 3) % mod != 1u) { eval_done[g] = E.wave_epoch; return; }      // evaluated this wave
    for (int a = 0; a < G::A; ++a) E.nn_policy[(size_t)g * G::A + a] = __builtin_nanf("");
    E.nn_value[g] = __builtin_nanf("");
    if (g % 3 == 0) atomic_add(fault, (int32_t)1);
}

template <class G> GAZ_KERNEL k_release(DevParams<G> E, const int32_t* moves) {
    const int g = block_id();
    if (g >= E.n_games || lane_id() != 0) return;
    GameState<G>& gs = E.games[g];
    if (gs.phase == PH_WAIT_HOST || (gs.phase == PH_IDLE && moves && moves[g] >= 0)) { gs.host_move = moves ? moves[g] : -1; gs.phase = PH_APPLY; }
}

// dense copies of the last MCTS.run result of every game (engine_get_root_stats)
template <class G> GAZ_KERNEL k_gather_root(DevParams<G> E, uint32_t* oN, float* oW, float* oP, float* oPol, uint32_t* oRV,
                                             float* oQ, int32_t* oChosen, int32_t* oPhase, int32_t* oPending) {
    using RL = RecLayout<G>;
    const int g = block_id();
    if (g >= E.n_games) return;
    const GameState<G>& gs = E.games[g];
    const uint8_t* rec = E.recs + (size_t)g * RL::SIZE;
    const int ply = gs.n_hist < G::MAXT ? gs.n_hist : G::MAXT - 1;
    for (int a = lane_id(); a < G::A; a += WAVE) {
        oN[(size_t)g * G::A + a] = reinterpret_cast<const uint32_t*>(rec + RL::OFF_N)[(size_t)ply * G::A + a];
        oW[(size_t)g * G::A + a] = reinterpret_cast<const float*>(rec + RL::OFF_W)[(size_t)ply * G::A + a];
        oP[(size_t)g * G::A + a] = reinterpret_cast<const float*>(rec + RL::OFF_P)[(size_t)ply * G::A + a];
        oPol[(size_t)g * G::A + a] = reinterpret_cast<const float*>(rec + RL::OFF_POL)[(size_t)ply * G::A + a];
    }
    if (lane_id() == 0) {
        oRV[g] = reinterpret_cast<const uint32_t*>(rec + RL::OFF_RV)[ply];
        oQ[g] = reinterpret_cast<const float*>(rec + RL::OFF_Q)[ply];
        oChosen[g] = gs.chosen; oPhase[g] = gs.phase; oPending[g] = gs.pend_kind;
    }
}

template <class G> GAZ_KERNEL k_count(DevParams<G> E, int32_t* out) {   // out[0] = games still searching, out[1] = pending evals
    const int g = block_id();
    if (g >= E.n_games || lane_id() != 0) return;
    const GameState<G>& gs = E.games[g];
    if (gs.phase != PH_WAIT_HOST && gs.phase != PH_HALT && gs.phase != PH_IDLE) atomic_add(&out[0], 1);
    if (gs.pend_kind != PEND_NONE) atomic_add(&out[1], 1);
    atomic_add(reinterpret_cast<unsigned long long*>(out + 2), (unsigned long long)gs.n_evals);
    atomic_add(reinterpret_cast<unsigned long long*>(out + 4), (unsigned long long)gs.n_sims);
    atomic_add(reinterpret_cast<unsigned long long*>(out + 6), (unsigned long long)gs.n_plies);
    atomic_add(reinterpret_cast<unsigned long long*>(out + 8), (unsigned long long)gs.n_hits);
}

// ------------------------------------------------------------------------------------------ engine
struct gaz_engine {
    gaz_engine_config cfg;
    std::string err;
    int grouped = 0;                                 // > 1: this engine is one of that many game groups of a GroupEngine (launch-shape defaults differ)
    virtual void note_grouped() {}
    virtual ~gaz_engine() {}
    int fail(const std::string& m) { err = m; return 1; }
    virtual int init() = 0;
    virtual int load_weights(const gaz_tensor* t, int n) = 0;
    virtual int reset_games(const int32_t* slots, int n) = 0;
    virtual int run_move(int32_t* n_waiting) = 0;
    virtual int get_root_stats(uint32_t*, float*, float*, float*, uint32_t*, float*, int32_t*, int32_t*) = 0;
    virtual int apply_moves(const int32_t* moves) = 0;
    virtual int run_waves(int n) = 0;
    virtual int wave_begin() = 0;
    virtual int wave_end() = 0;
    virtual int batch_ptrs(void**, void**, void**) = 0;
    virtual int read_batch(int8_t*, int32_t*) = 0;
    virtual int write_outputs(const float*, const float*) = 0;
    virtual int evaluate(const int8_t*, int, float*, float*, int, double*) = 0;
    virtual int record_layout(gaz_record_layout*) = 0;
    virtual int drain(void* out, int max_records, int32_t* n_out) = 0;
    virtual int get_stats(uint64_t out[16]) = 0;
    virtual int synchronize() = 0;
    virtual int timing_reset(int enable) = 0;
    virtual int timing_get(double*, double*, double*, int64_t*, int64_t*) = 0;
    virtual int dominant(char*, int, double*) = 0;
    virtual int set_position(int, const int32_t*, int) = 0;
    virtual int set_search_params(int, int) = 0;
    virtual int stop_search(int) = 0;
    virtual int start_search() = 0;
    virtual int set_hyperparams(const gaz_search_hyperparams*) = 0;
    virtual int read_head_features(int, float*, float*, int32_t*, int32_t*) = 0;
    virtual int set_fused_wave(int) = 0;
    virtual int debug_fused_fault(int) = 0;
    virtual int read_positions(int32_t*, uint8_t*, int) = 0;
    virtual int repack(int32_t*, int32_t*) = 0;
    virtual int probe_rules(const int32_t*, const int32_t*, int, int, int8_t*, uint8_t*, int32_t*, int8_t*, int32_t*, const float*, float*) = 0;
};

template <class G> struct EngineT : gaz_engine {
    using RL = RecLayout<G>;
    DevParams<G> E;
    hipStream_t stream = 0;
    Evaluator* eval = nullptr;
    // dense scratch for get_root_stats / counters
    uint32_t* dN = nullptr; float* dW = nullptr; float* dP = nullptr; float* dPol = nullptr; uint32_t* dRV = nullptr;
    float* dQ = nullptr; int32_t* dChosen = nullptr; int32_t* dPhase = nullptr; int32_t* dPending = nullptr;
    int32_t* dCount = nullptr; int32_t* dMoves = nullptr; int32_t* dSlots = nullptr;
    uint32_t ring_consumed = 0;
    bool timing = false;
    std::vector<hipEvent_t> ev;      // every timing event (owned)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_tree, ev_eval;   // brackets around tree steps / evaluator passes
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_fused;           // brackets around fused tree + trunk launches
    int64_t n_waves_total = 0, n_waves_timed = 0;
    static constexpr int TIMING_STRIDE = 8;
    std::vector<void*> allocs;

    template <class T> int dalloc(T** p, size_t n) {
        void* q = nullptr;
        HIP_OK(hipMalloc(&q, n * sizeof(T)));
        HIP_OK(hipMemset(q, 0, n * sizeof(T)));
        allocs.push_back(q); *p = (T*)q; return 0;
    }

    ~EngineT() override {
        hipStreamSynchronize(stream);
        if (E.prof) {                               // GAZ_TREE_PROF=1: phase cycles of the PUCT kernel, summed over games, to stderr
            std::vector<unsigned long long> h((size_t)E.n_games * 8);
            hipMemcpy(h.data(), E.prof, h.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < h.size(); ++i) t[i % 8] += h[i];
            const char* nm[7] = {"consume", "select / descent", "expand_pre", "cache_probe", "expand_post(hit)", "terminal backup", "whole launch"};
            fprintf(stderr, "[tree prof] launches/game %.0f\n", (double)t[7] / E.n_games);
            for (int k = 0; k < 7; ++k) fprintf(stderr, "[tree prof] %-18s %8.0f cycles / game-launch (%.1f %%)\n", nm[k], (double)t[k] / (double)t[7], 100.0 * t[k] / t[6]);
        }
        delete eval;
        for (void* p : allocs) hipFree(p);
        for (hipEvent_t e : ev) hipEventDestroy(e);
        if (pipeline_ready) {
            hipStreamSynchronize(tstream); hipStreamSynchronize(hstream);
            for (int p = 0; p < 2; ++p) for (int i = 0; i < n_grp; ++i) { hipEventDestroy(ev_tree_done[p][i]); hipEventDestroy(ev_trunk_done[p][i]); hipEventDestroy(ev_heads_done[p][i]); }
            hipEventDestroy(ev_join); hipStreamDestroy(tstream); hipStreamDestroy(hstream);
        }
        hipStreamDestroy(stream);
    }

    int init() override {
        HIP_OK(hipSetDevice(cfg.device));
        HIP_OK(hipStreamCreate(&stream));
        memset(&E, 0, sizeof(E));
        const int n = cfg.n_games;
        if (n <= 0) return fail("n_games must be positive");
        const bool gumbel = cfg.search == GAZ_SEARCH_GUMBEL;
        if (cfg.search != GAZ_SEARCH_PUCT && !gumbel) return fail("unknown search id");
        E.c_visit = cfg.c_visit; E.c_scale = cfg.c_scale; E.gumbel_m = cfg.gumbel_m; E.g_stablemax = cfg.gumbel_stablemax; E.fast_find_win = cfg.fast_find_win;
        if (gumbel && (cfg.gumbel_m < 2 || cfg.run_iterations < 1)) return fail("Gumbel search needs m >= 2 and run_iterations >= 1");
        E.node_bytes = gumbel ? gumbel_node_bytes<G>() : NodeLayout<G>::SIZE;
        // re-root compaction: needed where a whole-game arena does not fit (Gomoku: 4.2 KB records); 0 = auto
        E.compact = gumbel ? 0 : (cfg.compact_trees == 0 ? (G::ID == GAME_GMK ? 1 : 0) : (cfg.compact_trees > 0 ? 1 : 0));
        E.n_games = n; n_eff = n; E.run_iterations = cfg.run_iterations; E.max_actions = cfg.max_actions;
        if (cfg.max_actions > G::MAXT || cfg.max_actions <= 0) return fail("max_actions out of range for this game");
        E.explore_first = cfg.num_explore_actions_first; E.explore_second = cfg.num_explore_actions_second;
        E.create_new_root = cfg.create_new_root; E.sync_moves = cfg.sync_moves; E.use_dirichlet = cfg.use_dirichlet;
        int npt = cfg.nodes_per_tree;
        if (npt <= 0 && gumbel) npt = 2 * ((cfg.move_time_limit > 0.0 && 3 * G::A > cfg.run_iterations ? 3 * G::A : cfg.run_iterations) + cfg.gumbel_m) + 3 * G::A + 64;   // fresh tree every move
        if (npt <= 0 && E.compact) {   // per half: kept subtree + one run; generous bound, ERR_ARENA_FULL if a game exceeds it
            const int its = cfg.run_iterations < 3 * G::A ? 3 * G::A : cfg.run_iterations;
            npt = 4 * its + 3 * G::A + 64;
        }
        if (npt <= 0) {   // a tree lives for the whole game and gains <= 1 record per simulation of its own moves
            const int own_moves = (cfg.max_actions + 1) / 2 + 1;
            int its = cfg.run_iterations < 3 * G::A ? 3 * G::A : cfg.run_iterations;
            npt = own_moves * (its + 2) + 64;
        }
        E.nodes_per_tree = npt;
        E.ring_cap = cfg.ring_capacity; E.single_tree = cfg.single_tree;
        E.tau = norm_tau(cfg.tau); E.no_gumbel_noise = cfg.no_gumbel_noise; E.first_game_seq = cfg.first_game_seq;
        if (!(cfg.move_time_limit >= 0.0) || cfg.move_time_limit > 1e6) return fail("move_time_limit must be in [0, 1e6] seconds");
        if (cfg.move_time_limit > 0.0 && cfg.sync_moves) return fail("move_time_limit is for continuous self-play; the per-move API has gaz_engine_stop_search");
        E.move_time_ticks = cfg.move_time_limit > 0.0 ? (uint64_t)(cfg.move_time_limit * 1e8) + 1 : 0;
        if (cfg.games_budget < 0) return fail("games_budget must be >= 0");
        if (cfg.games_budget > 0 && cfg.sync_moves) return fail("games_budget needs continuous self-play (sync_moves = 0)");
        E.games_budget = cfg.games_budget;
        if (cfg.n_opening < 0 || cfg.n_opening > 8) return fail("at most 8 opening_actions");
        E.n_opening = cfg.n_opening;
        for (int i = 0; i < cfg.n_opening; ++i) {
            if (cfg.opening_actions[i] < 0 || cfg.opening_actions[i] >= G::A) return fail("opening action out of range");
            E.opening_actions[i] = cfg.opening_actions[i]; E.opening_weights[i] = cfg.opening_weights[i];
        }
        // measured: 32 -> 4 cuts the kernel tail 0.167 -> 0.067 ms; with the evaluation cache a hit is such a simulation too and 8 pays
        // (small boards; a Gomoku simulation is ten times as long and 8 only stretches the launch)
        // (round 2) with the fused tree + trunk launch the tail of the tree step hides behind the trunk: 12 pays there (69.6 k vs 67.9 k)
        // (round 3) with the completion queue and games that yield inside the phase loop, the tree step's tail no longer holds the trunk back, so more
        // evaluation-free simulations per launch only mean fewer rows without a request: Connect4 PUCT 4: 57.5 k, 8: 59.1 - 60.1 k, 12: 58.7 k;
        // Gumbel 4: 317 k, 8: 331 k, 16: 337 k; Gomoku 4: 2096, 16: 2177, 32: 2192 positions/s (tools/sweep_mts.sh, one box each)
        const bool resnet = cfg.evaluator == GAZ_EVAL_RESNET;
        const bool fusable = G::ID == GAME_C4 && cfg.search == GAZ_SEARCH_PUCT && resnet;
        const int fused_default = !resnet ? 0 : (G::ID == GAME_C4 ? (gumbel ? (cfg.eval_cache_log2 > 0 ? 0 : 16) : (cfg.eval_cache_log2 > 0 ? 12 : 8))
                                                                  : (G::ID == GAME_GMK && !gumbel && cfg.eval_cache_log2 == 0 ? 32 : 0));
        E.max_tree_sims = cfg.max_tree_sims_per_wave > 0 ? cfg.max_tree_sims_per_wave
                        : (fused_default ? fused_default : ((cfg.eval_cache_log2 > 0 && G::A <= 64) ? (fusable ? 12 : 8) : 4));
        E.c_init = cfg.c_puct_init; E.c_base = cfg.c_puct_base;
        E.alpha = (double)(float)cfg.dirichlet_alpha;     // alpha * np.ones_like(float32 policy) is float32 (MCTS.py:244-245)
        E.eps = cfg.dirichlet_epsilon; E.one_minus_eps = (float)(1.0 - cfg.dirichlet_epsilon);
        E.key0 = (uint32_t)cfg.seed; E.key1 = (uint32_t)(cfg.seed >> 32); E.slot_offset = cfg.slot_offset;
        const size_t arena_bytes = (size_t)n * 2 * (E.compact ? 2 : 1) * (size_t)npt * (size_t)E.node_bytes;
        void* a = nullptr;
        HIP_OK(hipMalloc(&a, arena_bytes));               // not zeroed: every record is written before it is read
        allocs.push_back(a); E.arena = (uint8_t*)a;
        if (dalloc(&E.trees, (size_t)n * 2)) return 1;
        if (dalloc(&E.games, (size_t)n)) return 1;
        if (dalloc(&E.paths, (size_t)n * PathCap<G>::V)) return 1;
        if (gumbel) { GumbelState<G>* gp = nullptr; if (dalloc(&gp, (size_t)n)) return 1; E.gstate = gp; }
        if (dalloc(&E.recs, (size_t)n * RL::SIZE)) return 1;
        if (dalloc(&E.ring, (size_t)(cfg.ring_capacity > 0 ? cfg.ring_capacity : 1) * RL::SIZE)) return 1;
        if (dalloc(&E.ring_head, 4)) return 1;
        if (dalloc(&E.nn_in, (size_t)n * G::HW * G::C + 64)) return 1;
        if (dalloc(&E.nn_policy, (size_t)n * G::A + 64)) return 1;
        if (dalloc(&E.nn_value, (size_t)n + 64)) return 1;
        if (cfg.eval_cache_log2 < 0 || cfg.eval_cache_log2 > 28) return fail("eval_cache_log2 out of range (0 = off, at most 28)");
        if (cfg.eval_cache_log2 > 0) {
            const size_t slots = (size_t)1 << cfg.eval_cache_log2;
            E.cache_stride = CacheLayout<G>::SIZE; E.cache_mask = (uint32_t)(slots - 1); E.cache_epoch = 1;
            if (dalloc(&E.cache, slots * (size_t)E.cache_stride)) return 1;      // zeroed: tag 0 never matches (tags are odd)
            if (dalloc(&E.cache_lock, slots)) return 1;
        }
        if (getenv("GAZ_TREE_PROF") && atoi(getenv("GAZ_TREE_PROF"))) { if (dalloc(&E.prof, (size_t)n * 8)) return 1; }
        if (!gumbel) {
            double* tb = nullptr;
            if (dalloc(&tb, (size_t)2 * PUCT_TABLE_N)) return 1;
            E.puct_table = tb;
            if (init_puct_table()) return 1;
        }
        if (dalloc(&E.stats, 8)) return 1;
        if (dalloc(&E.error, 4)) return 1;
        if (dalloc(&dN, (size_t)n * G::A) || dalloc(&dW, (size_t)n * G::A) || dalloc(&dP, (size_t)n * G::A) ||
            dalloc(&dPol, (size_t)n * G::A) || dalloc(&dRV, n) || dalloc(&dQ, n) || dalloc(&dChosen, n) ||
            dalloc(&dPhase, n) || dalloc(&dPending, n) || dalloc(&dCount, 16) || dalloc(&dMoves, (size_t)n + G::MAXT) || dalloc(&dSlots, n)) return 1;
        std::string e2;
        eval = make_evaluator(cfg, G::H, G::W, G::C, G::A, &e2);
        if (!eval && cfg.evaluator != GAZ_EVAL_EXTERNAL) return fail("evaluator: " + e2);
        GAZ_LAUNCH(k_init_games<G>, n, WAVE, stream, E, (int)cfg.first_game_seq);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }

    void note_grouped() override { if (eval) eval->set_shared_chip(this->grouped > 1); }

    int check_device_error() {
        int32_t code = 0;
        HIP_OK(hipMemcpy(&code, E.error, sizeof(code), hipMemcpyDeviceToHost));
        if (code) {
            const char* names[] = {"", "tree arena full (raise nodes_per_tree)", "root not fully expanded at move end",
                                   "selection path longer than the game can last", "state-machine loop guard", "PUCT picked an un-poppable child"};
            return fail(std::string("device error: ") + (code > 0 && code < 6 ? names[code] : "unknown"));
        }
        return 0;
    }

    int load_weights(const gaz_tensor* t, int n) override {
        if (!eval) return fail("no evaluator to load weights into");
        std::string m;
        if (eval->load(t, n, stream, &m)) return fail("load_weights: " + m);
        if (E.cache) {                              // cached outputs belong to the previous weights
            HIP_OK(hipMemsetAsync(E.cache, 0, ((size_t)E.cache_mask + 1) * (size_t)E.cache_stride, stream));
            HIP_OK(hipStreamSynchronize(stream));
        }
        return 0;
    }

    int reset_games(const int32_t* slots, int n) override {
        if (slots && n_eff != E.n_games) return fail("reset_games(slots): not after gaz_engine_repack (a physical slot no longer identifies a game; reset all games instead)");
        if (!slots) n_eff = E.n_games;               // every slot restarts: the launches cover all of them again
        if (slots) {
            if (n > E.n_games) return fail("reset_games: too many slots");
            HIP_OK(hipMemcpyAsync(dSlots, slots, sizeof(int32_t) * n, hipMemcpyHostToDevice, stream));
            GAZ_LAUNCH(k_reset_games<G>, n, WAVE, stream, E, (const int32_t*)dSlots, n);
        } else {
            GAZ_LAUNCH(k_reset_games<G>, E.n_games, WAVE, stream, E, (const int32_t*)nullptr, E.n_games);
        }
        HIP_OK(hipGetLastError());
        return 0;
    }

    hipEvent_t new_event() { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); return e; }
    static constexpr size_t MAX_TIMING_EVENTS = 1 << 16;            // a timed run of any length holds at most this many events

    // every tree launch carries the wave's epoch, and the marks of the previous wave's evaluator pass if that pass made any (a fused launch, or
    // the test hook): DevParams::eval_done
    bool prev_marked = false;                        // the wave before this one wrote eval_done
    DevParams<G> wave_params() { DevParams<G> P = E; P.wave_epoch = ++fuse_epoch; P.eval_done = prev_marked ? d_evaldone : nullptr; return P; }
    void launch_wave(hipStream_t st, int g0, int g1) { const DevParams<G> P = wave_params(); prev_marked = false; launch_wave(st, g0, g1, P); }   // (callers follow with a full evaluator pass)
    void launch_wave(hipStream_t st, int g0, int g1, const DevParams<G>& E) {
        // the small boards run four games per wavefront (PuctVariant: same records, 16-lane teams); GAZ_TREE_TEAMS=0 -> one per wave
        typedef typename PuctVariant<G>::type GP;
        if (cfg.search == GAZ_SEARCH_GUMBEL) {
            static const int gteams_env = getenv("GAZ_TREE_TEAMS") ? atoi(getenv("GAZ_TREE_TEAMS")) : -1;
            constexpr int GPER = WAVE / GP::TEAM;
            // measured (8192 Connect4 games, n = 32, m = 7): the team build is bit-exact but SLOWER, 225 vs 185 us per launch (sequential
            // halving keeps the four games of a wave in different phases) — opt-in only (GAZ_TREE_TEAMS=1)
            if (GPER > 1 && gteams_env == 1) GAZ_LAUNCH(k_wave_gumbel_teams<GP>, (g1 - g0 + GPER - 1) / GPER, WAVE, st, *reinterpret_cast<const DevParams<GP>*>(&E), g0, g1);
            else GAZ_LAUNCH(k_wave_gumbel<G>, g1 - g0, WAVE, st, E, g0, g1);
            return;
        }
        // measured (4096 Connect4 games): 66 vs 71 us per launch; with the evaluation cache (up to 8 evaluation-free simulations per
        // game and launch, and a wave is as slow as its slowest team) one game per wave stays faster: 60.1 k vs 56.2 k positions/s
        static const int teams_env = getenv("GAZ_TREE_TEAMS") ? atoi(getenv("GAZ_TREE_TEAMS")) : -1;
        const bool teams = teams_env >= 0 ? teams_env != 0 : E.cache == nullptr;
        constexpr int PER = WAVE / GP::TEAM;
        if (PER > 1 && teams) GAZ_LAUNCH(k_wave_teams<GP>, (g1 - g0 + PER - 1) / PER, WAVE, st, *reinterpret_cast<const DevParams<GP>*>(&E), g0, g1);
        else GAZ_LAUNCH(k_wave<G>, g1 - g0, WAVE, st, E, g0, g1);
    }
    void launch_wave() { launch_wave(stream, 0, n_eff); }

    // ---- fused tree + trunk launch (resnet.hip k_wave_trunk): Connect4 PUCT with the whole-trunk ResNet evaluator, no evaluation
    // cache (its probe reads rows other teams are writing).  GAZ_FUSE_WAVE=0 -> separate launches.
    // The hand-over inside the launch is BOUNDED (trunk.hpp TrunkArgs::spin_ticks): HIP promises no dispatch order, so a trunk workgroup whose
    // games' tree block is not resident may not wait forever.  A workgroup that gives up counts itself in d_fault and — unlike the others —
    // does not mark its boards in d_evaldone; their games keep their requests pending (no result changes), and poll_fuse_fault() — at every host
    // synchronisation point — switches this engine to separate launches for good.
    uint32_t* d_done = nullptr; uint32_t* d_evaldone = nullptr; int32_t* d_fault = nullptr;
    unsigned long long* d_queue = nullptr;           // completion queue of the fused launch (DevParams::done_queue); GAZ_FUSE_QUEUE=0 -> a flag per board
    uint32_t fuse_epoch = 0; int fuse_state = -1;     // -1 not decided, 0 off, 1 on
    uint64_t fuse_faults = 0;                        // trunk workgroups that gave up waiting, over the engine's life (get_stats [13])
    unsigned debug_fault_mod = 0;                    // test hook: gaz_engine_debug_fused_fault
    static constexpr unsigned SPIN_TICKS = 2000000;  // 20 ms of the 100-MHz wall clock; a tree step takes ~0.1 ms
    int poll_fuse_fault() {                          // stream must be idle
        if (!d_fault) return 0;
        int32_t n = 0;
        HIP_OK(hipMemcpy(&n, d_fault, sizeof(n), hipMemcpyDeviceToHost));
        if (n) {
            fuse_faults += (uint64_t)n; n = 0; fuse_state = 0; debug_fault_mod = 0;      // (the test hook injects until the fault has been seen)
            fault_wave = waves_launched; fault_strikes++;
            HIP_OK(hipMemcpy(d_fault, &n, sizeof(n), hipMemcpyHostToDevice));
            fprintf(stderr, "[gaz_engine] fused tree + trunk launch: %llu trunk workgroup(s) gave up waiting for their games (dispatch order not as assumed); "
                            "falling back to separate launches%s\n", (unsigned long long)fuse_faults,
                    fault_strikes <= REARM_STRIKES ? " (the one-launch form is tried again after 20000 waves)" : " for good");
        }
        return 0;
    }
    // A give-up can be a transient — with two game groups, two launches in flight can for once hold each other's slots — and separate launches cost a
    // grouped engine a fifth of its rate: after REARM_WAVES waves without the one-launch form it is tried again, at most REARM_STRIKES times.
    int64_t waves_launched = 0, fault_wave = 0; int fault_strikes = 0;
    static constexpr int64_t REARM_WAVES = 20000; static constexpr int REARM_STRIKES = 2;
    bool can_fuse() {
        if (fuse_state == 0 && fault_strikes > 0 && fault_strikes <= REARM_STRIKES && d_queue_or_done() && waves_launched - fault_wave >= REARM_WAVES) fuse_state = 1;
        if (fuse_state >= 0) return fuse_state == 1;
        fuse_state = 0;
        static const bool off = getenv("GAZ_FUSE_WAVE") && atoi(getenv("GAZ_FUSE_WAVE")) == 0;
        typedef typename PuctVariant<G>::type GP;
        // with the evaluation cache too (measured 67.5 k vs 64.3 k positions/s): a team's probe reads its OWN row, and the table is
        // written by k_cache_insert between launches as before.  GAZ_FUSE_CACHE=0 -> separate launches when the cache is on
        static const bool with_cache = !(getenv("GAZ_FUSE_CACHE") && atoi(getenv("GAZ_FUSE_CACHE")) == 0);
        const bool gumbel = cfg.search == GAZ_SEARCH_GUMBEL;
        // round 3: Gomoku's PUCT search as well (one game per wavefront; compacting trees are fine: a game's state lives in HBM either way).  GAZ_FUSE_GOMOKU=0 -> separate
        const bool gmk = G::ID == GAME_GMK && !gumbel && !E.cache && !(getenv("GAZ_FUSE_GOMOKU") && atoi(getenv("GAZ_FUSE_GOMOKU")) == 0);
        const bool c4 = G::ID == GAME_C4 && WAVE / GP::TEAM >= 4 && !E.compact && !(E.cache && !with_cache);
        if (off || !(c4 || gmk) || !eval || !eval->supports_split()) return false;
        // round 3: the Gumbel search too (BASELINE configs[4]; its tree step is 17 % of a wave when launched separately).  GAZ_FUSE_GUMBEL=0 -> separate
        if (gumbel && ((getenv("GAZ_FUSE_GUMBEL") && atoi(getenv("GAZ_FUSE_GUMBEL")) == 0) || E.cache)) return false;
        if (!gumbel && getenv("GAZ_TREE_TEAMS") && atoi(getenv("GAZ_TREE_TEAMS")) == 0) return false;
        if (dalloc(&d_done, (size_t)E.n_games) || !ensure_skip_buffers()) return false;
        if (!(getenv("GAZ_FUSE_QUEUE") && atoi(getenv("GAZ_FUSE_QUEUE")) == 0) && dalloc(&d_queue, (size_t)E.n_games + 64)) return false;
        fuse_state = 1;
        return true;
    }

    bool d_queue_or_done() const { return d_done != nullptr; }      // the fused launch's buffers exist: it had been running before the fault
    bool fuse_enabled = true;
    int set_fused_wave(int on) override { fuse_enabled = on != 0; return 0; }
    bool ensure_skip_buffers() {
        if (d_evaldone) return true;
        if (dalloc(&d_evaldone, (size_t)E.n_games) || dalloc(&d_fault, 4)) return false;
        return true;
    }
    int debug_fused_fault(int mod) override {
        if (mod > 0 && !ensure_skip_buffers()) return 1;
        debug_fault_mod = mod > 0 ? (unsigned)mod : 0u;
        return 0;
    }

    // ---- repack (continuous self-play with a games_budget): towards the end of a generation more and more slots have played their
    // last game and halted, but every wave still steps and EVALUATES all n_games rows.  repack() moves the games that still run into
    // the lowest slots and shrinks every later launch (tree step, evaluator, cache insert) to them: the tail of a generation costs
    // what its live games cost.  Results do not change: a game carries its identity (slot_id) and all its state with it.
    int n_eff = 0;                                   // launches cover physical slots [0, n_eff)
    int32_t* dMoveList = nullptr;
    int repack(int32_t* n_active_out, int32_t* n_eff_out) override {
        if (E.sync_moves) return fail("repack needs continuous self-play (sync_moves = 0)");
        HIP_OK(hipStreamSynchronize(stream));
        if (pipeline_ready) { HIP_OK(hipStreamSynchronize(tstream)); HIP_OK(hipStreamSynchronize(hstream)); }
        GAZ_LAUNCH(k_gather_root<G>, E.n_games, WAVE, stream, E, dN, dW, dP, dPol, dRV, dQ, dChosen, dPhase, dPending);
        std::vector<int32_t> ph(E.n_games);
        HIP_OK(hipMemcpyAsync(ph.data(), dPhase, (size_t)E.n_games * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        int n_active = 0;
        for (int g = 0; g < n_eff; ++g) n_active += ph[g] != PH_HALT;
        for (int g = n_eff; g < E.n_games; ++g) if (ph[g] != PH_HALT) return fail("repack: a slot beyond the live range is running");
        if (n_active_out) *n_active_out = n_active;
        const int n_new = n_active > 0 ? n_active : 1;
        std::vector<int32_t> mv;
        int hole = 0;
        for (int g = n_eff - 1; g >= n_new; --g) {
            if (ph[g] == PH_HALT) continue;
            while (hole < n_new && ph[hole] != PH_HALT) ++hole;
            if (hole >= n_new) return fail("repack: internal error (no hole left)");
            mv.push_back(g); mv.push_back(hole); ph[hole] = ph[g]; ph[g] = PH_HALT;
        }
        if (!mv.empty()) {
            if (!dMoveList && dalloc(&dMoveList, (size_t)2 * E.n_games)) return 1;
            HIP_OK(hipMemcpyAsync(dMoveList, mv.data(), mv.size() * 4, hipMemcpyHostToDevice, stream));
            const size_t slot_bytes = (size_t)2 * (E.compact ? 2 : 1) * (size_t)E.nodes_per_tree * (size_t)E.node_bytes;
            const size_t gbytes = cfg.search == GAZ_SEARCH_GUMBEL ? sizeof(GumbelState<G>) : 0;
#ifdef GAZ_HOST_EMU
            DevParams<G> Em = E; Em.eval_done = d_evaldone;              // a game's "evaluated last wave" mark travels with it
            GAZ_LAUNCH(k_move_slots<G>, (int)(mv.size() / 2), 1, stream, Em, (const int32_t*)dMoveList, (int)(mv.size() / 2), slot_bytes, gbytes);
#else
            DevParams<G> Em = E; Em.eval_done = d_evaldone;              // a game's "evaluated last wave" mark travels with it
            GAZ_LAUNCH(k_move_slots<G>, (int)(mv.size() / 2), 256, stream, Em, (const int32_t*)dMoveList, (int)(mv.size() / 2), slot_bytes, gbytes);
#endif
            HIP_OK(hipGetLastError());
            HIP_OK(hipStreamSynchronize(stream));
        }
        n_eff = n_new;
        if (n_eff_out) *n_eff_out = n_eff;
        return 0;
    }

    int one_wave(bool with_eval) {
        ++waves_launched;
        if (with_eval && eval && fuse_enabled && can_fuse()) {
            const bool timing = this->timing && n_waves_total % TIMING_STRIDE == 0 && ev.size() + 4 <= MAX_TIMING_EVENTS;
            // diagnostic: GAZ_FUSED_STAMPS=<file>[:n] -> wall-clock stamps of every block of the n-th fused launch (tools/fused_timeline.py)
            static const char* stamp_spec = getenv("GAZ_FUSED_STAMPS");
            static const long stamp_at = stamp_spec && strchr(stamp_spec, ':') ? atol(strchr(stamp_spec, ':') + 1) : 3000;
            unsigned long long* d_stamps = nullptr;
            const size_t stamp_blocks = (size_t)n_eff + 4096;              // >= tree blocks + trunk workgroups
            if (stamp_spec && n_waves_total == stamp_at && hipMalloc((void**)&d_stamps, stamp_blocks * 128 * 8) == hipSuccess) hipMemsetAsync(d_stamps, 0, stamp_blocks * 128 * 8, stream);
            const FuseHandoff ho{d_done, fuse_epoch + 1, d_evaldone, d_fault, SPIN_TICKS, debug_fault_mod, d_stamps, d_queue};
            const void* plan = eval->trunk_plan(E.nn_in, n_eff, 0, ho);
            if (plan) {
                hipEvent_t e0 = 0, e1 = 0, e2 = 0;
                if (timing) { e0 = new_event(); e1 = new_event(); e2 = new_event(); hipEventRecord(e0, stream); }
                DevParams<G> Ef = wave_params(); Ef.done_flag = d_done;
                if (eval->plan_uses_queue(plan)) {   // the trunk workgroups take queue entries: the tree blocks rank their games as they finish
                    typedef typename PuctVariant<G>::type GPq;
                    static const bool gumbel_teams = !(getenv("GAZ_FUSE_GUMBEL_TEAMS") && atoi(getenv("GAZ_FUSE_GUMBEL_TEAMS")) == 0);
                    // games per tree block = rounds x (4 waves x games per wavefront); rounds: measured, see DESIGN (GAZ_FUSE_TREE_ROUNDS overrides)
                    static const int rounds_env = getenv("GAZ_FUSE_TREE_ROUNDS") ? atoi(getenv("GAZ_FUSE_TREE_ROUNDS")) : 0;
                    // measured on one box (tools/sweep_rounds.sh): PUCT 1 round 56.6 k positions/s, 2 rounds 54.6 k, 4: 53.8 k; Gumbel (8192 games: every slot
                    // starts as a tree block) 1: 314 k, 2: 319 k, 4: 321 k, 8: 317 k
                    // Gomoku as one of two game groups (tools/sweep_groups5.sh): 2: 2372, 3: 2372, 4: 2384, 6: 2357, 8: 2330 positions/s
                    // Gumbel as one of two game groups of 4096 (tools/sweep_groups6.sh, sweep_groups7.sh, one box): 1: 341.3 k, 2: 344.2 k, 4: 321.4 k, 8: 247.5 k (one batch: 323.9 k)
                    const int rounds = rounds_env > 0 ? rounds_env : (G::ID == GAME_GMK ? (this->grouped > 1 ? 4 : 8) : (cfg.search == GAZ_SEARCH_GUMBEL ? (this->grouped > 1 ? 2 : 4) : 1));
                    const int gpb = G::ID == GAME_GMK ? rounds * 8 : rounds * 4 * ((cfg.search == GAZ_SEARCH_GUMBEL && !gumbel_teams) ? 1 : WAVE / GPq::TEAM);
                    Ef.done_queue = d_queue; Ef.queue_gpb = gpb; Ef.queue_nfull = n_eff / gpb; Ef.queue_rem = n_eff % gpb;
                }
                Ef.handoff_release = G::ID == GAME_GMK;
                const bool launched = G::ID == GAME_GMK ? launch_wave_trunk_gmk(stream, &Ef, 0, n_eff, plan)
                                    : cfg.search == GAZ_SEARCH_GUMBEL ? launch_wave_trunk_c4_gumbel(stream, &Ef, 0, n_eff, plan) : launch_wave_trunk_c4(stream, &Ef, 0, n_eff, plan);
                if (!launched) {                     // no fused kernel for this trunk variant: separate launches, for good
                    fuse_state = 0; --fuse_epoch;
                    return one_wave(with_eval);
                }
                prev_marked = true;                  // the trunk workgroups of this launch mark the boards they evaluate
                if (d_stamps) {
                    std::vector<unsigned long long> hst(stamp_blocks * 128);
                    hipStreamSynchronize(stream);
                    hipMemcpy(hst.data(), d_stamps, hst.size() * 8, hipMemcpyDeviceToHost); hipFree(d_stamps);
                    std::string path(stamp_spec); if (path.find(':') != std::string::npos) path = path.substr(0, path.find(':'));
                    if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(hst.data(), 8, hst.size(), f); fclose(f); }
                }
                if (timing) hipEventRecord(e1, stream);
                eval->forward_heads(stream, E.nn_policy, E.nn_value, n_eff, 0);
                if (E.cache) { Ef.eval_done = d_evaldone; GAZ_LAUNCH(k_cache_insert<G>, n_eff, WAVE, stream, Ef, 0, n_eff); E.cache_epoch++; }
                if (timing) {       // the fused kernel is booked as evaluator time; tree time is what it hides
                    hipEventRecord(e2, stream); ev_eval.push_back({e0, e2}); n_waves_timed++;
                    ev_fused.push_back({e0, e1});
                }
                n_waves_total++;
                return 0;
            }
        }
        // timing brackets on every TIMING_STRIDE-th wave only: an event record is a barrier packet of its own (~5 us between two
        // kernels; five of them per wave were 4 % of the wave they measured)
        const bool timing = this->timing && n_waves_total % TIMING_STRIDE == 0 && ev.size() + 4 <= MAX_TIMING_EVENTS;
        hipEvent_t e0 = 0, e1 = 0, e2 = 0;
        if (timing) { e0 = new_event(); e1 = new_event(); e2 = new_event(); hipEventRecord(e0, stream); }
        DevParams<G> P = wave_params();
        launch_wave(stream, 0, n_eff, P);
        prev_marked = false;                         // the separately launched evaluator pass covers every row
        if (timing) hipEventRecord(e1, stream);
        if (with_eval && eval) eval->forward(stream, E.nn_in, E.nn_policy, E.nn_value, n_eff, timing);
        if (timing) { hipEventRecord(e2, stream); ev_tree.push_back({e0, e1}); ev_eval.push_back({e1, e2}); n_waves_timed++; }
        P.eval_done = nullptr;
        if (with_eval && eval && debug_fault_mod && d_evaldone) {      // test hook: this wave behaves as a fused launch with workgroups that gave up
            GAZ_LAUNCH(k_debug_skip<G>, n_eff, WAVE, stream, P, d_evaldone, n_eff, debug_fault_mod, d_fault);
            prev_marked = true; P.eval_done = d_evaldone;
        }
        if (with_eval && eval && E.cache) { GAZ_LAUNCH(k_cache_insert<G>, n_eff, WAVE, stream, P, 0, n_eff); E.cache_epoch++; }
        n_waves_total++;
        return 0;
    }

    // ---- group pipeline (free-running self-play only).  The games are split into K groups; stream `stream` carries nothing but the
    // trunk kernels trunk(g0, k), trunk(g1, k), ..., trunk(g0, k + 1) back to back — the MFMA kernel is the only thing on the critical
    // path — while the latency-bound rest runs behind it on two other streams: heads(g, k) (Dense-1 + tail, `hstream`) and the next
    // tree step tree(g, k + 1) (`tstream`) of a group overlap the trunk launches of the OTHER groups.  Group sizes are whole rounds
    // of the trunk kernel's tiles (2 workgroups x CUs x 3 boards = 1536 Connect4 boards; the last group may be the cheaper 2-board
    // tiles), so splitting costs no MFMA round.  Per-game results do not depend on the grouping: rows of a batch are independent.
    hipStream_t tstream = 0, hstream = 0;
    static constexpr int MAXGRP = 8;
    int n_grp = 0, grp0[MAXGRP + 1] = {};
    hipEvent_t ev_tree_done[2][MAXGRP] = {}, ev_trunk_done[2][MAXGRP] = {}, ev_heads_done[2][MAXGRP] = {}, ev_join = 0;
    bool pipeline_ready = false;
    bool can_pipeline() {
        static const char* spec = getenv("GAZ_PIPELINE");
        if (!spec || !*spec || atoi(spec) == 0) return false;
        // (the evaluation cache is read by tree kernels and written by k_cache_insert: they must not run concurrently)
        if (E.sync_moves || !eval || !eval->supports_split() || E.n_games < 1024 || E.cache || n_eff != E.n_games) return false;
        if (!pipeline_ready) {
            // GAZ_PIPELINE=1: groups of one full trunk round (1536 boards), remainder last; GAZ_PIPELINE=a,b,c: explicit sizes
            std::vector<int> sizes;
            if (strchr(spec, ',')) { for (const char* p = spec; *p;) { sizes.push_back(atoi(p)); p = strchr(p, ','); if (!p) break; ++p; } }
            else { const int round = eval->round_rows(); for (int left = E.n_games; left > 0; left -= round) sizes.push_back(left < round ? left : round); }
            int sum = 0; for (int v : sizes) sum += v;
            if (sizes.size() < 2 || sizes.size() > (size_t)MAXGRP || sum != E.n_games) return false;
            for (int v : sizes) if (v <= 0) return false;
            n_grp = (int)sizes.size(); grp0[0] = 0;
            for (int i = 0; i < n_grp; ++i) grp0[i + 1] = grp0[i] + sizes[i];
            int lo = 0, hi = 0;
            hipDeviceGetStreamPriorityRange(&lo, &hi);                   // the small kernels should be dispatched ahead of queued trunk tiles
            if (hipStreamCreateWithPriority(&tstream, hipStreamNonBlocking, hi) != hipSuccess) return false;
            if (hipStreamCreateWithPriority(&hstream, hipStreamNonBlocking, hi) != hipSuccess) return false;
            for (int p = 0; p < 2; ++p) for (int i = 0; i < n_grp; ++i) {
                hipEventCreateWithFlags(&ev_tree_done[p][i], hipEventDisableTiming); hipEventCreateWithFlags(&ev_trunk_done[p][i], hipEventDisableTiming);
                hipEventCreateWithFlags(&ev_heads_done[p][i], hipEventDisableTiming);
            }
            hipEventCreateWithFlags(&ev_join, hipEventDisableTiming);
            pipeline_ready = true;
        }
        return true;
    }
    int run_waves_pipelined(int n) {
        const int in_row = G::HW * G::C;
        hipEventRecord(ev_join, stream);                                 // the side streams start after everything queued on `stream`
        hipStreamWaitEvent(tstream, ev_join, 0); hipStreamWaitEvent(hstream, ev_join, 0);
        for (int k = 0; k < n; ++k) {
            const int cur = k & 1, prev = cur ^ 1;
            const bool timed = timing && n_waves_total % TIMING_STRIDE == 0 && ev.size() + 2 * (size_t)n_grp <= MAX_TIMING_EVENTS;
            const DevParams<G> P = wave_params(); prev_marked = false;    // one epoch per wave, whatever the number of groups
            for (int g = 0; g < n_grp; ++g) {
                const int g0 = grp0[g], g1 = grp0[g + 1];
                if (k > 0) hipStreamWaitEvent(tstream, ev_heads_done[prev][g], 0);      // this group's previous evaluation
                hipEvent_t t0 = 0, t1 = 0;
                if (timed) { t0 = new_event(); t1 = new_event(); hipEventRecord(t0, tstream); }
                launch_wave(tstream, g0, g1, P);
                if (timed) { hipEventRecord(t1, tstream); ev_tree.push_back({t0, t1}); }
                hipEventRecord(ev_tree_done[cur][g], tstream);
                hipStreamWaitEvent(stream, ev_tree_done[cur][g], 0);
                eval->forward_trunk(stream, E.nn_in + (size_t)g0 * in_row, g1 - g0, timed, g0);
                hipEventRecord(ev_trunk_done[cur][g], stream);
                hipStreamWaitEvent(hstream, ev_trunk_done[cur][g], 0);
                eval->forward_heads(hstream, E.nn_policy + (size_t)g0 * G::A, E.nn_value + g0, g1 - g0, g0);
                hipEventRecord(ev_heads_done[cur][g], hstream);
            }
            n_waves_total++; if (timed) n_waves_timed++;
        }
        // join: `stream` ends after every side-stream kernel, so synchronize() / stats reads on it see a quiescent engine
        hipEventRecord(ev_join, tstream); hipStreamWaitEvent(stream, ev_join, 0);
        for (int g = 0; g < n_grp; ++g) hipStreamWaitEvent(stream, ev_heads_done[(n - 1) & 1][g], 0);
        HIP_OK(hipGetLastError());
        return 0;
    }

    int counts(int32_t out[10]) {
        HIP_OK(hipMemsetAsync(dCount, 0, 10 * sizeof(int32_t), stream));
        GAZ_LAUNCH(k_count<G>, E.n_games, WAVE, stream, E, dCount);
        HIP_OK(hipMemcpyAsync(out, dCount, 10 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }

    int run_move(int32_t* n_waiting) override {
        if (!E.sync_moves) return fail("run_move needs sync_moves = 1");
        if (!eval) return fail("run_move needs a built-in evaluator (use wave_begin/wave_end with GAZ_EVAL_EXTERNAL)");
        if (!eval->ready()) return fail("run_move: evaluator weights not loaded (gaz_engine_load_weights)");
        // a move needs at most iter_limit + 2 evaluations (both roots); poll the device every 16 waves
        // PUCT: a move needs at most iter_limit + 2 evaluations; Gumbel can overshoot its budget (vpc >= 1 per survivor)
        const int max_waves = cfg.search == GAZ_SEARCH_GUMBEL ? 4 * (E.run_iterations + 3 * G::A) + 64
                                                             : (E.run_iterations < 3 * G::A ? 3 * G::A : E.run_iterations) + 8;
        int32_t c[10];
        for (int w = 0; w < max_waves + 16; w += 16) {
            for (int i = 0; i < 16; ++i) one_wave(true);
            if (counts(c)) return 1;
            if (check_device_error()) return 1;
            if (c[0] == 0) break;
        }
        if (c[0] != 0) return fail("run_move: games still searching after the wave budget");
        if (n_waiting) *n_waiting = E.n_games;
        return 0;
    }

    int get_root_stats(uint32_t* oN, float* oW, float* oP, float* oPol, uint32_t* oRV, float* oQ, int32_t* oChosen,
                       int32_t* oPhase) override {
        GAZ_LAUNCH(k_gather_root<G>, E.n_games, WAVE, stream, E, dN, dW, dP, dPol, dRV, dQ, dChosen, dPhase, dPending);
        HIP_OK(hipGetLastError());
        const size_t na = (size_t)E.n_games * G::A, n = E.n_games;
        if (oN) HIP_OK(hipMemcpyAsync(oN, dN, na * 4, hipMemcpyDeviceToHost, stream));
        if (oW) HIP_OK(hipMemcpyAsync(oW, dW, na * 4, hipMemcpyDeviceToHost, stream));
        if (oP) HIP_OK(hipMemcpyAsync(oP, dP, na * 4, hipMemcpyDeviceToHost, stream));
        if (oPol) HIP_OK(hipMemcpyAsync(oPol, dPol, na * 4, hipMemcpyDeviceToHost, stream));
        if (oRV) HIP_OK(hipMemcpyAsync(oRV, dRV, n * 4, hipMemcpyDeviceToHost, stream));
        if (oQ) HIP_OK(hipMemcpyAsync(oQ, dQ, n * 4, hipMemcpyDeviceToHost, stream));
        if (oChosen) HIP_OK(hipMemcpyAsync(oChosen, dChosen, n * 4, hipMemcpyDeviceToHost, stream));
        if (oPhase) HIP_OK(hipMemcpyAsync(oPhase, dPhase, n * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }

    int apply_moves(const int32_t* moves) override {
        if (!E.sync_moves) return fail("apply_moves needs sync_moves = 1");
        if (moves) HIP_OK(hipMemcpyAsync(dMoves, moves, sizeof(int32_t) * E.n_games, hipMemcpyHostToDevice, stream));
        GAZ_LAUNCH(k_release<G>, E.n_games, WAVE, stream, E, moves ? (const int32_t*)dMoves : (const int32_t*)nullptr);
        HIP_OK(hipGetLastError());
        // run the APPLY phase (do_action, win check, prune) up to the next evaluation request
        launch_wave();
        if (eval) eval->forward(stream, E.nn_in, E.nn_policy, E.nn_value, E.n_games, false);
        if (eval && E.cache) { GAZ_LAUNCH(k_cache_insert<G>, E.n_games, WAVE, stream, E, 0, E.n_games); E.cache_epoch++; }
        HIP_OK(hipGetLastError());
        return check_device_error();
    }

    int run_waves(int n) override {
        if (!eval) return fail("run_waves needs a built-in evaluator");
        if (!eval->ready()) return fail("run_waves: evaluator weights not loaded (gaz_engine_load_weights)");
        if (can_pipeline()) return run_waves_pipelined(n);
        for (int i = 0; i < n; ++i) one_wave(true);
        HIP_OK(hipGetLastError());
        return 0;
    }

    int wave_begin() override { one_wave(false); HIP_OK(hipGetLastError()); return 0; }
    int wave_end() override { return 0; }   // the next wave_begin consumes the outputs; nothing to do here

    int batch_ptrs(void** a, void** b, void** c) override {
        if (a) *a = E.nn_in; if (b) *b = E.nn_policy; if (c) *c = E.nn_value; return 0;
    }
    int read_batch(int8_t* inputs, int32_t* pending) override {
        if (inputs) HIP_OK(hipMemcpyAsync(inputs, E.nn_in, (size_t)E.n_games * G::HW * G::C, hipMemcpyDeviceToHost, stream));
        if (pending) {
            GAZ_LAUNCH(k_gather_root<G>, E.n_games, WAVE, stream, E, dN, dW, dP, dPol, dRV, dQ, dChosen, dPhase, dPending);
            HIP_OK(hipMemcpyAsync(pending, dPending, (size_t)E.n_games * 4, hipMemcpyDeviceToHost, stream));
        }
        HIP_OK(hipStreamSynchronize(stream));
        return check_device_error();
    }
    int write_outputs(const float* policy, const float* value) override {
        HIP_OK(hipMemcpyAsync(E.nn_policy, policy, (size_t)E.n_games * G::A * 4, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(E.nn_value, value, (size_t)E.n_games * 4, hipMemcpyHostToDevice, stream));
        if (E.cache) { GAZ_LAUNCH(k_cache_insert<G>, E.n_games, WAVE, stream, E, 0, E.n_games); E.cache_epoch++; }
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }

    int evaluate(const int8_t* in, int n, float* pol, float* val, int repeats, double* ms) override {
        if (!eval) return fail("evaluate: no built-in evaluator");
        if (!eval->ready()) return fail("evaluate: evaluator weights not loaded");
        if (n <= 0 || n > E.n_games) return fail("evaluate: n must be in [1, n_games]");
        HIP_OK(hipMemcpyAsync(E.nn_in, in, (size_t)n * G::HW * G::C, hipMemcpyHostToDevice, stream));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        eval->forward(stream, E.nn_in, E.nn_policy, E.nn_value, n, false);      // warm-up / the measured result
        hipEventRecord(e0, stream);
        for (int r = 0; r < repeats; ++r) eval->forward(stream, E.nn_in, E.nn_policy, E.nn_value, n, false);
        hipEventRecord(e1, stream);
        HIP_OK(hipMemcpyAsync(pol, E.nn_policy, (size_t)n * G::A * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(val, E.nn_value, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        float t = 0; hipEventElapsedTime(&t, e0, e1); hipEventDestroy(e0); hipEventDestroy(e1);
        if (ms) *ms = repeats > 0 ? (double)t / repeats : 0.0;
        HIP_OK(hipGetLastError());
        return 0;
    }

    int read_head_features(int n, float* p, float* v, int32_t* p_row, int32_t* v_row) override {
        const float *dp = nullptr, *dv = nullptr; int pr = 0, vr = 0;
        if (!eval || !eval->head_features(&dp, &dv, &pr, &vr)) return fail("read_head_features: this evaluator keeps no head features");
        if (n < 0 || n > E.n_games) return fail("read_head_features: n must be in [0, n_games]");
        if (p_row) *p_row = pr; if (v_row) *v_row = vr;
        HIP_OK(hipStreamSynchronize(stream));
        if (p && n) HIP_OK(hipMemcpy(p, dp, (size_t)n * pr * 4, hipMemcpyDeviceToHost));
        if (v && n) HIP_OK(hipMemcpy(v, dv, (size_t)n * vr * 4, hipMemcpyDeviceToHost));
        return 0;
    }

    int record_layout(gaz_record_layout* o) override {
        o->record_bytes = RL::SIZE; o->max_T = G::MAXT; o->A = G::A; o->t_pad = G::TPAD;
        o->off_hdr = RL::OFF_HDR; o->off_actions = RL::OFF_ACT; o->off_q = RL::OFF_Q; o->off_root_visits = RL::OFF_RV;
        o->off_evals = RL::OFF_EV; o->off_policy = RL::OFF_POL; o->off_N = RL::OFF_N; o->off_W = RL::OFF_W; o->off_P = RL::OFF_P;
        return 0;
    }

    int drain(void* out, int max_records, int32_t* n_out) override {
        *n_out = 0;
        if (E.ring_cap <= 0) return 0;
        HIP_OK(hipStreamSynchronize(stream));
        if (poll_fuse_fault()) return 1;            // a host synchronisation point like the others: a caller that only ever drains must still get the fallback
        uint32_t head[2];
        HIP_OK(hipMemcpy(head, E.ring_head, sizeof(head), hipMemcpyDeviceToHost));
        uint32_t avail = head[0] - ring_consumed;
        if ((int)avail > max_records) avail = (uint32_t)max_records;
        for (uint32_t i = 0; i < avail; ++i) {
            const uint32_t slot = (ring_consumed + i) % (uint32_t)E.ring_cap;
            HIP_OK(hipMemcpy((uint8_t*)out + (size_t)i * RL::SIZE, E.ring + (size_t)slot * RL::SIZE, RL::SIZE, hipMemcpyDeviceToHost));
        }
        ring_consumed += avail;
        HIP_OK(hipMemcpy(E.ring_head + 1, &ring_consumed, sizeof(uint32_t), hipMemcpyHostToDevice));
        *n_out = (int32_t)avail;
        return check_device_error();
    }

    int get_stats(uint64_t out[16]) override {
        for (int i = 0; i < 16; ++i) out[i] = 0;
        int32_t c[10];
        if (counts(c)) return 1;
        unsigned long long s[8];
        HIP_OK(hipMemcpy(s, E.stats, sizeof(s), hipMemcpyDeviceToHost));
        for (int i = 0; i < 6; ++i) out[i] = s[i];
        memcpy(&out[6], c + 2, 8); memcpy(&out[7], c + 4, 8); memcpy(&out[8], c + 6, 8);
        out[9] = (uint64_t)n_waves_total; memcpy(&out[10], c + 8, 8);
        out[11] = pipeline_ready ? (uint64_t)n_grp : 0;
        if (poll_fuse_fault()) return 1;             // counts() synchronised the stream
        out[12] = (fuse_state == 1 && fuse_enabled) ? 1 : 0;
        out[13] = fuse_faults;
        return check_device_error();
    }

    int synchronize() override { HIP_OK(hipStreamSynchronize(stream)); if (poll_fuse_fault()) return 1; return check_device_error(); }

    int timing_reset(int enable) override {
        HIP_OK(hipStreamSynchronize(stream));
        for (hipEvent_t e : ev) hipEventDestroy(e);
        ev.clear(); ev_tree.clear(); ev_eval.clear(); ev_fused.clear(); n_waves_total = 0; n_waves_timed = 0; timing = enable != 0;
        if (eval) eval->timing_reset();
        return 0;
    }
    int set_position(int slot, const int32_t* actions, int n) override {
        if (n_eff != E.n_games) return fail("set_position: not after gaz_engine_repack (a physical slot no longer identifies a game)");
        if (slot < 0 || slot >= E.n_games || n < 0 || n > G::MAXT) return fail("set_position: bad slot / history length");
        if (n > 0) HIP_OK(hipMemcpyAsync(dMoves, actions, sizeof(int32_t) * n, hipMemcpyHostToDevice, stream));   // dMoves holds >= MAXT? n_games ints
        GAZ_LAUNCH(k_set_position<G>, 1, WAVE, stream, E, slot, (const int32_t*)dMoves, n);
        HIP_OK(hipGetLastError());
        return 0;
    }
    uint8_t* dHist = nullptr;
    int read_positions(int32_t* n_hist, uint8_t* hist, int stride) override {
        if (!n_hist || !hist || stride < G::MAXT) return fail("read_positions: hist must hold at least max_T actions per slot");
        if (stride > G::TPAD) stride = G::TPAD;
        if (!dHist && dalloc(&dHist, (size_t)E.n_games * G::TPAD)) return 1;
        GAZ_LAUNCH(k_read_positions<G>, E.n_games, WAVE, stream, E, dPhase, dHist, stride);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(n_hist, dPhase, (size_t)E.n_games * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(hist, dHist, (size_t)E.n_games * stride, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }
    int start_search() override { GAZ_LAUNCH(k_start_search<G>, E.n_games, WAVE, stream, E); HIP_OK(hipGetLastError()); return 0; }
    int stop_search(int stop) override { E.stop_search = stop != 0; return 0; }
    // tau != 0 and tau <= 5e-3 -> 0 (MCTS.py:116-120,163-168); negative = Self_Play's schedule
    static double norm_tau(double tau) { return tau < 0.0 ? -1.0 : ((tau != 0.0 && tau <= 5e-3) ? 0.0 : tau); }
    int set_search_params(int run_iterations, int tau_mode) override {
        if (run_iterations > 0) E.run_iterations = run_iterations;
        E.tau = tau_mode < 0 ? -1.0 : (double)tau_mode;
        return 0;
    }
    int init_puct_table() {
        double* tb = const_cast<double*>(E.puct_table);
#ifdef GAZ_HOST_EMU
        GAZ_LAUNCH(k_init_puct_table<0>, 1, 1, stream, tb, E.c_init, E.c_base, PUCT_TABLE_N);
#else
        GAZ_LAUNCH(k_init_puct_table<0>, (PUCT_TABLE_N + 255) / 256, 256, stream, tb, E.c_init, E.c_base, PUCT_TABLE_N);
#endif
        HIP_OK(hipGetLastError());
        return 0;
    }
    // MCTS.update_hyperparams (MCTS.py:134-168: invalid values are ignored with a warning there, rejected here) and
    // MCTS_Gumbel.update_hyperparams (MCTS_Gumbel.py:186-210); kernels read DevParams per launch, so the next wave sees them
    int set_hyperparams(const gaz_search_hyperparams* hp) override {
        if (!hp || hp->struct_size != sizeof(gaz_search_hyperparams)) return fail("set_hyperparams: struct_size does not match this library's gaz_search_hyperparams");
        auto given = [](double v) { return v == v; };
        if (given(hp->c_puct_init) && hp->c_puct_init < 0.0) return fail("c_puct_init cannot be negative");
        if (given(hp->c_puct_base) && hp->c_puct_base <= 0.0) return fail("c_puct_base must be positive");
        if (given(hp->dirichlet_alpha) && hp->dirichlet_alpha <= 0.0) return fail("dirichlet_alpha must be positive");
        if (given(hp->dirichlet_epsilon) && (hp->dirichlet_epsilon < 0.0 || hp->dirichlet_epsilon >= 1.0)) return fail("dirichlet_epsilon must be in [0, 1)");
        if (hp->gumbel_m >= 0 && cfg.search == GAZ_SEARCH_GUMBEL && hp->gumbel_m < 2) return fail("Gumbel search needs m >= 2");
        // the node arena was sized at create time from run_iterations (and gumbel_m): a larger value now could only end in ERR_ARENA_FULL in
        // the middle of a search, so it is refused here with the remedy
        if (cfg.nodes_per_tree <= 0) {
            const int it = hp->run_iterations > 0 ? hp->run_iterations : E.run_iterations, m = hp->gumbel_m >= 0 ? hp->gumbel_m : E.gumbel_m;
            long long need = 0;
            if (cfg.search == GAZ_SEARCH_GUMBEL) need = 2LL * (it + m) + 3 * G::A + 64;
            else if (E.compact) need = 4LL * (it < 3 * G::A ? 3 * G::A : it) + 3 * G::A + 64;
            else need = (long long)((cfg.max_actions + 1) / 2 + 1) * ((it < 3 * G::A ? 3 * G::A : it) + 2) + 64;
            if (need > E.nodes_per_tree)
                return fail("set_hyperparams: run_iterations / gumbel_m beyond what the tree arena was sized for at create time (" + std::to_string(E.nodes_per_tree) +
                            " nodes per tree, " + std::to_string(need) + " needed): create the engine with the largest values, or give nodes_per_tree");
        }
        bool table = false;
        if (given(hp->c_puct_init)) { E.c_init = hp->c_puct_init; table = true; }
        if (given(hp->c_puct_base)) { E.c_base = hp->c_puct_base; table = true; }
        if (given(hp->dirichlet_alpha)) E.alpha = (double)(float)hp->dirichlet_alpha;
        if (given(hp->dirichlet_epsilon)) { E.eps = hp->dirichlet_epsilon; E.one_minus_eps = (float)(1.0 - hp->dirichlet_epsilon); }
        if (hp->use_dirichlet >= 0) E.use_dirichlet = hp->use_dirichlet != 0;
        if (given(hp->tau)) E.tau = norm_tau(hp->tau);
        if (hp->gumbel_m >= 0) E.gumbel_m = hp->gumbel_m;
        if (given(hp->c_visit)) E.c_visit = hp->c_visit;
        if (given(hp->c_scale)) E.c_scale = hp->c_scale;
        if (hp->run_iterations > 0) E.run_iterations = hp->run_iterations;
        if (table && E.puct_table) return init_puct_table();
        return 0;
    }
    int8_t* pr_board = nullptr; uint8_t* pr_legal = nullptr; int32_t* pr_winner = nullptr; int8_t* pr_input = nullptr; int32_t* pr_term = nullptr;
    int32_t* pr_actions = nullptr; int32_t* pr_n = nullptr; float* pr_pin = nullptr; float* pr_pout = nullptr; int pr_cap = 0, pr_stride = 0;
    int probe_rules(const int32_t* actions, const int32_t* n_actions, int n_pos, int stride, int8_t* o_board, uint8_t* o_legal,
                    int32_t* o_winner, int8_t* o_input, int32_t* o_term, const float* policy_in, float* o_policy) override {
        if (n_pos <= 0 || stride <= 0 || stride > G::MAXT || !actions || !n_actions) return fail("probe_rules: bad arguments");
        for (int p = 0; p < n_pos; ++p) {
            if (n_actions[p] < 0 || n_actions[p] > stride) return fail("probe_rules: history longer than stride");
            for (int i = 0; i < n_actions[p]; ++i) { const int a = actions[(size_t)p * stride + i]; if (a < 0 || a >= G::A) return fail("probe_rules: action out of range"); }
        }
        if (n_pos > pr_cap || stride > pr_stride) {                      // scratch grows, never shrinks (freed with the engine)
            const size_t n = (size_t)n_pos, st = (size_t)G::MAXT;
            if (dalloc(&pr_board, n * G::HW) || dalloc(&pr_legal, n * G::A) || dalloc(&pr_winner, n) || dalloc(&pr_input, n * G::HW * G::C) ||
                dalloc(&pr_term, n * G::A) || dalloc(&pr_actions, n * st) || dalloc(&pr_n, n) || dalloc(&pr_pin, n * G::A) || dalloc(&pr_pout, n * G::A)) return 1;
            pr_cap = n_pos; pr_stride = G::MAXT;
        }
        const size_t n = (size_t)n_pos;
        HIP_OK(hipMemcpyAsync(pr_actions, actions, n * stride * 4, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(pr_n, n_actions, n * 4, hipMemcpyHostToDevice, stream));
        const bool pol = policy_in && o_policy;
        if (pol) HIP_OK(hipMemcpyAsync(pr_pin, policy_in, n * G::A * 4, hipMemcpyHostToDevice, stream));
        GAZ_LAUNCH(k_probe_rules<G>, n_pos, WAVE, stream, E, (const int32_t*)pr_actions, (const int32_t*)pr_n, n_pos, stride, pr_board, pr_legal,
                   pr_winner, pr_input, pr_term, pol ? (const float*)pr_pin : (const float*)nullptr, pol ? pr_pout : (float*)nullptr);
        HIP_OK(hipGetLastError());
        if (o_board) HIP_OK(hipMemcpyAsync(o_board, pr_board, n * G::HW, hipMemcpyDeviceToHost, stream));
        if (o_legal) HIP_OK(hipMemcpyAsync(o_legal, pr_legal, n * G::A, hipMemcpyDeviceToHost, stream));
        if (o_winner) HIP_OK(hipMemcpyAsync(o_winner, pr_winner, n * 4, hipMemcpyDeviceToHost, stream));
        if (o_input) HIP_OK(hipMemcpyAsync(o_input, pr_input, n * G::HW * G::C, hipMemcpyDeviceToHost, stream));
        if (o_term) HIP_OK(hipMemcpyAsync(o_term, pr_term, n * G::A * 4, hipMemcpyDeviceToHost, stream));
        if (pol) HIP_OK(hipMemcpyAsync(o_policy, pr_pout, n * G::A * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        return 0;
    }
    int dominant(char* name, int cap, double* flops) override {
        // positions per evaluator launch; the pipelined run launches the trunk kernel once per group: price the MEAN launch (the
        // timing sums all launches, so FLOPs per launch x launches = FLOPs of the waves either way)
        const int n_launch = can_pipeline() ? E.n_games / n_grp : E.n_games;
        double f = 0; const char* k = eval ? eval->dominant_kernel(n_launch, &f) : "";
        if (eval && can_pipeline()) { double fa = 0; eval->dominant_kernel(E.n_games, &fa); f = fa / n_grp; }
        std::string label = k;
        if (fuse_state == 1 && fuse_enabled) label = "k_wave_trunk = " + label + " FUSED with the " + std::string(cfg.search == GAZ_SEARCH_GUMBEL ? "Gumbel" : "PUCT") + " tree step of the same wave (one launch: tree blocks first, trunk workgroups start on the "
                                     "boards whose games are done; the launch duration therefore includes the part of the tree step it could not hide)";
        if (name && cap > 0) { strncpy(name, label.c_str(), cap - 1); name[cap - 1] = 0; }
        if (flops) *flops = f;
        return 0;
    }
    int timing_get(double* ms_tree, double* ms_eval, double* ms_dom, int64_t* n_dom, int64_t* n_waves) override {
        HIP_OK(hipStreamSynchronize(stream));
        double t = 0, e = 0;
        for (auto& pr : ev_tree) { float a = 0; hipEventElapsedTime(&a, pr.first, pr.second); t += a; }
        for (auto& pr : ev_eval) { float b = 0; hipEventElapsedTime(&b, pr.first, pr.second); e += b; }
        if (ms_tree) *ms_tree = t; if (ms_eval) *ms_eval = e;
        double d = 0; int64_t nd = 0;
        if (eval) eval->timing_get(&d, &nd);
        for (auto& pr : ev_fused) { float a = 0; hipEventElapsedTime(&a, pr.first, pr.second); d += a; nd++; }
        if (ms_dom) *ms_dom = d; if (n_dom) *n_dom = nd; if (n_waves) *n_waves = n_waves_timed;
        return 0;
    }
};

// ------------------------------------------------------------------------------------------ game groups
// gaz_engine_config::game_groups: the games of ONE engine handle as K independent engines (groups of consecutive slots), each with its own
// stream, batch and fused tree + trunk launch, stepped alternately by run_waves.  Nothing of a game depends on how games are grouped — RNG streams
// and records are keyed by the GLOBAL slot (slot_offset + slot), rows of an evaluator batch are independent bit for bit — so every game is the one
// a single engine plays (tests: records equal by (slot, game_seq)); only the order in which finished games reach the ring differs.
// Why: a wave of one engine is tree step -> trunk -> heads in sequence; in the one-launch form the trunk workgroups still wait ~60 us for their
// first boards, the last round of tiles leaves slots idle and the heads (Dense-1 + tail, ~27 us) run on a nearly empty chip.  With TWO launches
// of half the games in flight, the other group's trunk tiles fill all of that: Connect4 headline config, steady state, one box: 7.84 M -> 8.60 M
// evaluations/s with 2048 | 2048 (2560 | 1536: 8.14 M; 3072 | 1024: 7.19 M; three groups 7.73 M; four 7.1 M;
// 8-lane teams = half the tree blocks: +0.5 %, within noise, not kept).  Gomoku 2048 games -> 2 x 1024: +6 % (+8.6 % with four tree rounds per block).
// Gumbel 8192 -> 2 x 4096 with two tree rounds per block: +6 %.  auto = 2 for Connect4 PUCT + ResNet from 3072 games, Gomoku PUCT + ResNet from
// 2048 games, Connect4 Gumbel + ResNet from 6144 games.
static gaz_engine* make_single_engine(const gaz_engine_config& cfg, std::string* err) {
    gaz_engine* h = nullptr;
    switch (cfg.game) {
        case GAZ_GAME_TICTACTOE: h = new EngineT<Game<GAME_TTT>>(); break;
        case GAZ_GAME_CONNECT4: h = new EngineT<Game<GAME_C4>>(); break;
        case GAZ_GAME_GOMOKU: h = new EngineT<Game<GAME_GMK>>(); break;
        default: *err = "unknown game id"; return nullptr;
    }
    h->cfg = cfg; h->cfg.game_groups = 1;
    if (h->init()) { *err = h->err; delete h; return nullptr; }
    return h;
}

struct GroupEngine : gaz_engine {
    std::vector<gaz_engine*> kid;
    std::vector<int> first;                          // first[c] = first slot of group c, first[K] = n_games
    gaz_record_layout lay{};
    int drain_from = 0;
    ~GroupEngine() override { for (gaz_engine* k : kid) delete k; }
    int K() const { return (int)kid.size(); }
    int size_of(int c) const { return first[c + 1] - first[c]; }
    int up(gaz_engine* k, int rc) { if (rc) err = k->err; return rc; }
    template <class F> int each(F f) { for (int c = 0; c < K(); ++c) if (up(kid[c], f(kid[c], c))) return 1; return 0; }
    int group_of(int slot) const { int c = 0; while (c + 1 < K() && slot >= first[c + 1]) ++c; return c; }

    int init() override {
        const int n = cfg.n_games, k = cfg.game_groups;
        if (k < 2 || k > n) return fail("game_groups must be in [1, n_games]");
        if (cfg.sync_moves || cfg.evaluator == GAZ_EVAL_EXTERNAL) return fail("game_groups > 1 needs continuous self-play (sync_moves = 0) with a built-in evaluator");
        if (cfg.games_budget > 0 && cfg.games_budget < n) return fail("game_groups > 1 with a games_budget below n_games: a group could be left without a game (use game_groups = 1)");
        first.assign(1, 0);
        for (int c = 0; c < k; ++c) first.push_back(first.back() + n / k + (c < n % k ? 1 : 0));
        for (int c = 0; c < k; ++c) {
            gaz_engine_config cc = cfg;
            cc.n_games = size_of(c); cc.slot_offset = cfg.slot_offset + (uint32_t)first[c];
            if (cfg.ring_capacity > 0) cc.ring_capacity = (int32_t)(((int64_t)cfg.ring_capacity * cc.n_games + n - 1) / n);
            // with another group's trunk tiles filling the chip, the tail of a group's tree step costs less than the rows that carry no request:
            // two groups, headline config, one box (tools/sweep_groups.sh): 8: 66.8 k, 12: 67.5 k, 16: 67.4 k positions/s (one batch: 8, see EngineT::init)
            if (cfg.max_tree_sims_per_wave == 0 && cfg.game == GAZ_GAME_CONNECT4 && cfg.search == GAZ_SEARCH_PUCT && cfg.evaluator == GAZ_EVAL_RESNET) cc.max_tree_sims_per_wave = 12;
            if (cfg.games_budget > 0) {
                // one engine: slot g plays its k-th game iff k n + g < budget, i.e. floor(budget / n) games and one more in the first budget % n slots;
                // the group's own rule (k n_c + g_c < budget_c) gives exactly those games with this budget
                const int64_t whole = cfg.games_budget / n, extra = cfg.games_budget % n - first[c];
                cc.games_budget = whole * cc.n_games + (extra < 0 ? 0 : (extra > cc.n_games ? cc.n_games : extra));
            }
            std::string e;
            gaz_engine* h = make_single_engine(cc, &e);
            if (!h) return fail("game group " + std::to_string(c) + ": " + e);
            h->grouped = k; h->note_grouped();
            kid.push_back(h);
        }
        return kid[0]->record_layout(&lay);
    }
    int load_weights(const gaz_tensor* t, int n) override { return each([&](gaz_engine* k, int) { return k->load_weights(t, n); }); }
    int reset_games(const int32_t* slots, int n) override {
        if (!slots) return each([&](gaz_engine* k, int) { return k->reset_games(nullptr, 0); });
        std::vector<std::vector<int32_t>> per(K());
        for (int i = 0; i < n; ++i) {
            if (slots[i] < 0 || slots[i] >= cfg.n_games) return fail("reset_games: slot out of range");
            const int c = group_of(slots[i]); per[c].push_back(slots[i] - first[c]);
        }
        return each([&](gaz_engine* k, int c) { return per[c].empty() ? 0 : k->reset_games(per[c].data(), (int)per[c].size()); });
    }
    int run_move(int32_t* w) override { return up(kid[0], kid[0]->run_move(w)); }             // (sync_moves = 0: the group's own refusal)
    int apply_moves(const int32_t* m) override { return up(kid[0], kid[0]->apply_moves(m)); }
    int get_root_stats(uint32_t* N, float* W, float* P, float* pol, uint32_t* rv, float* q, int32_t* chosen, int32_t* phase) override {
        return each([&](gaz_engine* k, int c) {
            const size_t a = (size_t)first[c] * lay.A, g = (size_t)first[c];
            return k->get_root_stats(N ? N + a : N, W ? W + a : W, P ? P + a : P, pol ? pol + a : pol, rv ? rv + g : rv, q ? q + g : q, chosen ? chosen + g : chosen, phase ? phase + g : phase);
        });
    }
    int run_waves(int n) override {                  // wave by wave, group by group: every group's stream always holds work
        for (int i = 0; i < n; ++i) if (each([&](gaz_engine* k, int) { return k->run_waves(1); })) return 1;
        return 0;
    }
    int no_batch() { return fail("game groups hold one evaluator batch each: the wave_begin / batch API needs game_groups = 1"); }
    int wave_begin() override { return no_batch(); }
    int wave_end() override { return no_batch(); }
    int batch_ptrs(void**, void**, void**) override { return no_batch(); }
    int read_batch(int8_t*, int32_t*) override { return no_batch(); }
    int write_outputs(const float*, const float*) override { return no_batch(); }
    // rows [first[c], first[c + 1]) go through group c's evaluator (and stay in ITS head-feature buffers: read_head_features below); ms: the sum
    int evaluate(const int8_t* in, int n, float* p, float* v, int rep, double* ms) override {
        if (n < 1 || n > cfg.n_games) return fail("evaluate: n must be in [1, n_games]");
        const size_t row = cfg.game == GAZ_GAME_TICTACTOE ? Game<GAME_TTT>::HW * Game<GAME_TTT>::C : (cfg.game == GAZ_GAME_CONNECT4 ? Game<GAME_C4>::HW * Game<GAME_C4>::C : Game<GAME_GMK>::HW * Game<GAME_GMK>::C);
        double total = 0;
        if (each([&](gaz_engine* k, int c) {
                const int nc = n - first[c] < 0 ? 0 : (n - first[c] > size_of(c) ? size_of(c) : n - first[c]);
                double m = 0;
                if (nc && k->evaluate(in + (size_t)first[c] * row, nc, p ? p + (size_t)first[c] * lay.A : p, v ? v + first[c] : v, rep, &m)) return 1;
                total += m; return 0;
            })) return 1;
        if (ms) *ms = total;
        return 0;
    }
    int record_layout(gaz_record_layout* o) override { *o = lay; return 0; }
    int drain(void* out, int max_records, int32_t* n_out) override {
        int total = 0;
        for (int i = 0; i < K() && total < max_records; ++i) {       // start with another group every call: no ring starves behind a small max_records
            gaz_engine* k = kid[(drain_from + i) % K()];
            int32_t got = 0;
            if (up(k, k->drain((uint8_t*)out + (size_t)total * lay.record_bytes, max_records - total, &got))) return 1;
            total += got;
        }
        drain_from = (drain_from + 1) % K();
        *n_out = total;
        return 0;
    }
    int get_stats(uint64_t out[16]) override {
        for (int i = 0; i < 16; ++i) out[i] = 0;
        out[12] = 1;
        for (int c = 0; c < K(); ++c) {
            uint64_t s[16];
            if (up(kid[c], kid[c]->get_stats(s))) return 1;
            for (int i = 0; i < 9; ++i) out[i] += s[i];
            out[10] += s[10]; out[13] += s[13];
            if (s[9] > out[9]) out[9] = s[9];        // waves: every group has run the same number
            if (!s[12]) out[12] = 0;                 // one launch per wave: only if every group runs it
        }
        out[14] = (uint64_t)K();
        return 0;
    }
    int synchronize() override { return each([](gaz_engine* k, int) { return k->synchronize(); }); }
    int timing_reset(int en) override { return each([&](gaz_engine* k, int) { return k->timing_reset(en); }); }
    // sums over the groups: per wave (of all groups) the kernel time of every group's launches — they overlap in time, so the sum is NOT wall clock
    int timing_get(double* ms_tree, double* ms_eval, double* ms_dom, int64_t* n_dom, int64_t* n_waves) override {
        double t = 0, e = 0, d = 0; int64_t nd = 0, nw = 0;
        for (int c = 0; c < K(); ++c) {
            double a = 0, b = 0, x = 0; int64_t y = 0, z = 0;
            if (up(kid[c], kid[c]->timing_get(&a, &b, &x, &y, &z))) return 1;
            t += a; e += b; d += x; nd += y; if (z > nw) nw = z;
        }
        if (ms_tree) *ms_tree = t; if (ms_eval) *ms_eval = e; if (ms_dom) *ms_dom = d; if (n_dom) *n_dom = nd; if (n_waves) *n_waves = nw;
        return 0;
    }
    int dominant(char* name, int cap, double* flops) override { return up(kid[0], kid[0]->dominant(name, cap, flops)); }      // per launch of one group
    int set_position(int slot, const int32_t* a, int n) override {
        if (slot < 0 || slot >= cfg.n_games) return fail("set_position: bad slot / history length");
        const int c = group_of(slot);
        return up(kid[c], kid[c]->set_position(slot - first[c], a, n));
    }
    int set_search_params(int it, int tau) override { return each([&](gaz_engine* k, int) { return k->set_search_params(it, tau); }); }
    int stop_search(int s) override { return each([&](gaz_engine* k, int) { return k->stop_search(s); }); }
    int start_search() override { return each([](gaz_engine* k, int) { return k->start_search(); }); }
    int set_hyperparams(const gaz_search_hyperparams* hp) override { return each([&](gaz_engine* k, int) { return k->set_hyperparams(hp); }); }
    int read_head_features(int n, float* p, float* v, int32_t* p_row, int32_t* v_row) override {
        if (n < 0 || n > cfg.n_games) return fail("read_head_features: n must be in [0, n_games]");
        int32_t pr = 0, vr = 0;
        if (up(kid[0], kid[0]->read_head_features(0, nullptr, nullptr, &pr, &vr))) return 1;
        if (p_row) *p_row = pr; if (v_row) *v_row = vr;
        return each([&](gaz_engine* k, int c) {
            const int nc = n - first[c] < 0 ? 0 : (n - first[c] > size_of(c) ? size_of(c) : n - first[c]);
            return nc ? k->read_head_features(nc, p ? p + (size_t)first[c] * pr : p, v ? v + (size_t)first[c] * vr : v, nullptr, nullptr) : 0;
        });
    }
    int set_fused_wave(int on) override { return each([&](gaz_engine* k, int) { return k->set_fused_wave(on); }); }
    int debug_fused_fault(int mod) override { return each([&](gaz_engine* k, int) { return k->debug_fused_fault(mod); }); }
    int read_positions(int32_t* n_hist, uint8_t* hist, int stride) override {
        if (!n_hist || !hist || stride < lay.max_T) return fail("read_positions: hist must hold at least max_T actions per slot");
        const int st = stride > lay.t_pad ? lay.t_pad : stride;      // the row stride the groups write with
        return each([&](gaz_engine* k, int c) { return k->read_positions(n_hist + first[c], hist + (size_t)first[c] * st, stride); });
    }
    int repack(int32_t* n_active, int32_t* n_launch) override {      // every group packs its own live games; the sums are reported
        int32_t a = 0, l = 0;
        if (each([&](gaz_engine* k, int) { int32_t x = 0, y = 0; if (k->repack(&x, &y)) return 1; a += x; l += y; return 0; })) return 1;
        if (n_active) *n_active = a; if (n_launch) *n_launch = l;
        return 0;
    }
    int probe_rules(const int32_t* actions, const int32_t* n_actions, int n_pos, int stride, int8_t* b, uint8_t* l, int32_t* w, int8_t* in, int32_t* t,
                    const float* pin, float* pout) override { return up(kid[0], kid[0]->probe_rules(actions, n_actions, n_pos, stride, b, l, w, in, t, pin, pout)); }
};

// auto (game_groups = 0): two groups where that was measured to pay (see above); GAZ_GAME_GROUPS overrides the automatic choice only
static int choose_game_groups(const gaz_engine_config& c) {
    if (c.game_groups != 0) return c.game_groups;
    static const int env = getenv("GAZ_GAME_GROUPS") ? atoi(getenv("GAZ_GAME_GROUPS")) : 0;
    const bool able = !c.sync_moves && c.evaluator != GAZ_EVAL_EXTERNAL && !(c.games_budget > 0 && c.games_budget < c.n_games);
    if (env > 0) return (able && env <= c.n_games) ? env : 1;
    // (with the evaluation cache every group has a table of its own — nothing is shared between groups: 108.2 k -> 110.2 k positions/s)
    const bool pays = c.game == GAZ_GAME_CONNECT4 && c.search == GAZ_SEARCH_PUCT && c.evaluator == GAZ_EVAL_RESNET && c.net_blocks > 0 && c.n_games >= 3072;
    // Gomoku (2048 games, 10 blocks; the one-launch wave needs the cache off): 2195 -> 2330 positions/s, 2384 with four tree rounds per block
    // (EngineT::one_wave); from empty boards 1706 -> 1855.  Three groups 2225.  Its heads are a chain of small kernels (~300 us of a 3000-us wave)
    // that one batch runs on an idle chip
    const bool pays_gmk = c.game == GAZ_GAME_GOMOKU && c.search == GAZ_SEARCH_PUCT && c.evaluator == GAZ_EVAL_RESNET && c.net_blocks > 0 && c.eval_cache_log2 == 0 && c.n_games >= 2048;
    // Gumbel (8192 games -> 2 x 4096, two tree rounds per block instead of four): 323.9 k -> 344.1 k positions/s same box (with four rounds: -1 %)
    const bool pays_gumbel = c.game == GAZ_GAME_CONNECT4 && c.search == GAZ_SEARCH_GUMBEL && c.evaluator == GAZ_EVAL_RESNET && c.net_blocks > 0 && c.eval_cache_log2 == 0 && c.n_games >= 6144;
    return able && (pays || pays_gmk || pays_gumbel) ? 2 : 1;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

int gaz_engine_abi_version(void) { return GAZ_ENGINE_ABI_VERSION; }
int gaz_engine_config_size(void) { return (int)sizeof(gaz_engine_config); }

int gaz_engine_create(const gaz_engine_config* cfg, gaz_engine** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return 1; }
    *out = nullptr;
    if (cfg->struct_size != sizeof(gaz_engine_config)) {
        g_create_error = "gaz_engine_config.struct_size is " + std::to_string(cfg->struct_size) + ", this library's gaz_engine_config has " +
                         std::to_string(sizeof(gaz_engine_config)) + " bytes (ABI version " + std::to_string(GAZ_ENGINE_ABI_VERSION) +
                         "): rebuild the binding from include/gaz_engine.h";
        return 1;
    }
    if (cfg->game_groups < 0) { g_create_error = "game_groups must be >= 0"; return 1; }
    const int groups = choose_game_groups(*cfg);
    if (groups <= 1) {
        gaz_engine* h = make_single_engine(*cfg, &g_create_error);
        if (!h) return 1;
        *out = h;
        return 0;
    }
    GroupEngine* h = new GroupEngine();
    h->cfg = *cfg; h->cfg.game_groups = groups;
    if (h->init()) { g_create_error = h->err; delete h; return 1; }
    *out = h;
    return 0;
}
void gaz_engine_destroy(gaz_engine* h) { delete h; }
const char* gaz_engine_last_error(gaz_engine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }
int gaz_engine_load_weights(gaz_engine* h, const gaz_tensor* t, int32_t n) { return h->load_weights(t, n); }
int gaz_engine_reset_games(gaz_engine* h, const int32_t* slots, int32_t n) { return h->reset_games(slots, n); }
int gaz_engine_run_move(gaz_engine* h, int32_t* n_waiting) { return h->run_move(n_waiting); }
int gaz_engine_get_root_stats(gaz_engine* h, uint32_t* N, float* W, float* P, float* pol, uint32_t* rv, float* q, int32_t* chosen,
                              int32_t* phase) { return h->get_root_stats(N, W, P, pol, rv, q, chosen, phase); }
int gaz_engine_apply_moves(gaz_engine* h, const int32_t* moves) { return h->apply_moves(moves); }
int gaz_engine_run_waves(gaz_engine* h, int32_t n) { return h->run_waves(n); }
int gaz_engine_wave_begin(gaz_engine* h) { return h->wave_begin(); }
int gaz_engine_wave_end(gaz_engine* h) { return h->wave_end(); }
int gaz_engine_batch_ptrs(gaz_engine* h, void** a, void** b, void** c) { return h->batch_ptrs(a, b, c); }
int gaz_engine_read_batch(gaz_engine* h, int8_t* in, int32_t* pending) { return h->read_batch(in, pending); }
int gaz_engine_write_outputs(gaz_engine* h, const float* p, const float* v) { return h->write_outputs(p, v); }
int gaz_engine_evaluate(gaz_engine* h, const int8_t* in, int32_t n, float* p, float* v, int32_t repeats, double* ms) { return h->evaluate(in, n, p, v, repeats, ms); }
int gaz_engine_record_layout(gaz_engine* h, gaz_record_layout* o) { return h->record_layout(o); }
int gaz_engine_drain_finished(gaz_engine* h, void* out, int32_t max_records, int32_t* n_out) { return h->drain(out, max_records, n_out); }
int gaz_engine_get_stats(gaz_engine* h, uint64_t out[16]) { return h->get_stats(out); }
int gaz_engine_synchronize(gaz_engine* h) { return h->synchronize(); }
int gaz_engine_timing_reset(gaz_engine* h, int32_t enable) { return h->timing_reset(enable); }
int gaz_engine_set_position(gaz_engine* h, int32_t slot, const int32_t* actions, int32_t n) { return h->set_position(slot, actions, n); }
int gaz_engine_set_search_params(gaz_engine* h, int32_t run_iterations, int32_t tau_mode) { return h->set_search_params(run_iterations, tau_mode); }
int gaz_engine_stop_search(gaz_engine* h, int32_t stop) { return h->stop_search(stop); }
int gaz_engine_start_search(gaz_engine* h) { return h->start_search(); }
int gaz_engine_set_hyperparams(gaz_engine* h, const gaz_search_hyperparams* hp) { return h->set_hyperparams(hp); }
int gaz_engine_set_fused_wave(gaz_engine* h, int32_t on) { return h->set_fused_wave(on); }
int gaz_engine_debug_fused_fault(gaz_engine* h, int32_t mod) { return h->debug_fused_fault(mod); }
int gaz_engine_read_positions(gaz_engine* h, int32_t* n_hist, uint8_t* hist, int32_t stride) { return h->read_positions(n_hist, hist, stride); }
int gaz_engine_repack(gaz_engine* h, int32_t* n_active, int32_t* n_launch) { return h->repack(n_active, n_launch); }
int gaz_engine_read_head_features(gaz_engine* h, int32_t n, float* p, float* v, int32_t* p_row, int32_t* v_row) { return h->read_head_features(n, p, v, p_row, v_row); }
int gaz_engine_probe_rules(gaz_engine* h, const int32_t* actions, const int32_t* n_actions, int32_t n_positions, int32_t stride, int8_t* board,
                           uint8_t* legal, int32_t* winner, int8_t* input, int32_t* terminal, const float* policy_in, float* policy_out) {
    return h->probe_rules(actions, n_actions, n_positions, stride, board, legal, winner, input, terminal, policy_in, policy_out);
}
int gaz_engine_dominant_kernel(gaz_engine* h, char* name, int32_t cap, double* flops) { return h->dominant(name, cap, flops); }
int gaz_engine_timing_get(gaz_engine* h, double* a, double* b, double* c, int64_t* d, int64_t* e) { return h->timing_get(a, b, c, d, e); }

}  // extern "C"
