/*
 * gaz_engine.h — C ABI of the MI355X batched self-play engine (libgaz_engine.so).
 *
 * The reference (subtotechnoblade/Grok_Alpha_Zero, pure Python) has no FFI; its boundaries are the
 * duck-typed Python surfaces listed below.  The Python package grok_alpha_zero_amd/ re-presents those
 * surfaces (same names / arguments) on top of this ABI; INTEGRATION.md shows the ctypes stub a reference
 * maintainer would add.  Every entry point returns 0 on success, non-zero on failure
 * (gaz_engine_last_error gives the text); no exceptions, no torch types, caller-owned buffers,
 * one host thread per engine (= per GPU), HIP streams internal.
 *
 * Reference interface replaced by each entry point (file:line under /root/reference):
 *   gaz_engine_create          Self_Play.__init__ building MCTS x2 per game   Self_Play.py:16-69, MCTS.py:78-132
 *                              + run_self_play's worker / server start-up      Self_Play.py:259-363
 *   gaz_engine_load_weights    rt.InferenceSession(onnx_path) in the server    Client_Server.py:119-120
 *   gaz_engine_reset_games     self_play_task -> game_class()                  Self_Play.py:237-245
 *   gaz_engine_run_move        MCTS.run(iteration_limit) for every live game   MCTS.py:528-618 (Self_Play.py:97-106)
 *   gaz_engine_get_root_stats  the move_probs rows MCTS.run returns            MCTS.py:591-600
 *   gaz_engine_apply_moves     game.do_action + mcts1/2.prune_tree             Self_Play.py:142-157, MCTS.py:657-671
 *   gaz_engine_run_waves       the whole Self_Play.play loop, device resident  Self_Play.py:71-157
 *   gaz_engine_wave_begin/end  session.run(["policy","value"], {"inputs": x})  MCTS.py:224-235, Client_Server.py:28-55,162-217
 *   gaz_engine_drain_finished  the per-game arrays play() hands to HDF5        Self_Play.py:159-175
 *   gaz_engine_get_stats       file["game_stats"] u32[6]                       Self_Play.py:181-188
 *   gaz_engine_set_position    MCTS.__init__ attaching to a live game object   MCTS.py:100,132,296-313
 *   gaz_engine_read_positions  game.action_history of every game in progress      Guide.py:111-133 (the attribute MCTS reads at MCTS.py:297-313)
 *   gaz_engine_set_search_params  run(iteration_limit) / update_hyperparams(tau) MCTS.py:134-168,528
 *   gaz_engine_set_hyperparams    MCTS.update_hyperparams(c_puct_*, dirichlet_*, tau) MCTS.py:134-168;
 *                                 MCTS_Gumbel.update_hyperparams(m, c_visit, c_scale)  MCTS_Gumbel.py:186-210
 *   gaz_engine_probe_rules     the Game plugin's static *_MCTS functions          Guide.py:135-283, Game_Tester.py:297-405
 *   gaz_engine_stop_search        run(time_limit)                              MCTS.py:560-563
 *   gaz_engine_repack             finished workers no longer load the inference server  Self_Play.py:380-400
 *   gaz_engine_set_fused_wave     (scheduling switch; no reference counterpart: Client_Server.py's server loop is what it replaces)
 *   gaz_engine_debug_fused_fault  (test hook for that launch's bounded hand-over; what it replaces is the client's unbounded
 *                                  spin on the server's flag, Client_Server.py:42-55)
 *   gaz_engine_evaluate        sess.run on a stacked batch (evaluator probe)   Compute_Speed.py:40-63, Client_Server.py:199-206
 *   gaz_engine_read_head_features  intermediate tensors of that probe (numerics tests)  Connect4/Build_Model.py:41-47,62-66
 */
#ifndef GAZ_ENGINE_H
#define GAZ_ENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gaz_engine gaz_engine;

enum { GAZ_GAME_TICTACTOE = 0, GAZ_GAME_CONNECT4 = 1, GAZ_GAME_GOMOKU = 2 };
enum { GAZ_SEARCH_PUCT = 0, GAZ_SEARCH_GUMBEL = 1 };
enum { GAZ_EVAL_HASH = 0,      /* synthetic bit-reproducible evaluator (parity tests) */
       GAZ_EVAL_RESNET = 1,    /* the ResNet policy/value network, HIP MFMA kernels */
       GAZ_EVAL_EXTERNAL = 2   /* caller evaluates the batch between wave_begin / wave_end */ };

#define GAZ_ENGINE_ABI_VERSION 4   /* bumped whenever gaz_engine_config / gaz_search_hyperparams / an entry point changes */

typedef struct {
    uint32_t struct_size;         /* = sizeof(gaz_engine_config) of the header the caller was built against; gaz_engine_create
                                     rejects any other value (a stale binding would otherwise be read past its end) */
    int32_t game;                 /* GAZ_GAME_* */
    int32_t search;               /* GAZ_SEARCH_* */
    int32_t n_games;              /* concurrent games on this GPU */
    int32_t run_iterations;       /* iteration_limit passed to MCTS.run (Self_Play passes int(1.5*MCTS_iteration_limit)) */
    int32_t max_actions;          /* train_config["max_actions"] */
    int32_t num_explore_actions_first, num_explore_actions_second;
    double c_puct_init, c_puct_base;
    double dirichlet_alpha, dirichlet_epsilon;
    int32_t use_dirichlet;
    int32_t create_new_root;      /* train_config.get("create_new_root", False) */
    int32_t sync_moves;           /* 1: stop after each move for get_root_stats/apply_moves; 0: continuous self-play */
    int32_t nodes_per_tree;       /* arena capacity per (game, tree); 0 = default for the game */
    int32_t ring_capacity;        /* finished-game records kept for drain_finished; 0 = keep none */
    uint64_t seed;
    uint32_t slot_offset;         /* global slot of local game 0 (rank * n_games) */
    int32_t evaluator;            /* GAZ_EVAL_* */
    uint32_t hash_salt;
    int32_t device;               /* HIP device ordinal */
    /* ResNet (evaluator == GAZ_EVAL_RESNET): trunk of `net_blocks` pre-activation blocks x `net_filters` */
    int32_t net_blocks, net_filters;
    int32_t policy_is_logits;     /* policy head: 0 = softmax (float64, Build_Model.py:60), 1 = raw logits (Gumbel),
                                     2 = stablemax (Net/Stablemax.py:8-12, build_config["use_stablemax"]) */
    int32_t gumbel_m;             /* train_config["m"]: actions sampled in the first stage of sequential halving */
    double c_visit, c_scale;      /* train_config["c_visit"], ["c_scale"] (MCTS_Gumbel.py:160-161) */
    int32_t compact_trees;        /* re-root compaction of the tree arena: 0 = auto (on for Gomoku), 1 = on, -1 = off */
    int32_t single_tree;          /* 1: one tree plays both sides (MCTS used on its own: Connect4/play.py, Game_Tester.py:480-513) */
    int32_t n_opening;            /* train_config["opening_actions"] (Self_Play.py:130-140): up to 8 [action, weight] pairs */
    int32_t opening_actions[8];
    double opening_weights[8];
    int32_t max_tree_sims_per_wave; /* evaluation-free simulations a game may run per launch before it yields (0 = the configuration's measured default: 4 .. 32);
                                       scheduling only — results do not depend on it */
    int32_t eval_cache_log2;      /* on-device evaluation cache with 2^n entries, keyed by the encoded leaf state (replaces
                                     Session_Cache.Cache_Wrapper, Session_Cache.py:4-26 / Self_Play.py:234-236); 0 = off.
                                     A hit returns the bits the evaluator produced for the same input: results do not change */
    int32_t gumbel_stablemax;     /* 1: MCTS_Gumbel(activation_fn="stablemax") — build_config["use_stablemax"] (Self_Play.py:69):
                                     stablemax instead of softmax inside deterministic_selection (MCTS_Gumbel.py:144-148) */
    int32_t fast_find_win;        /* MCTS(fast_find_win=True) (MCTS.py:88,282-283; MCTS_Gumbel.py:313): a position with a winning
                                     move keeps only the first one (in legal-action order); Self_Play always passes False */
    int32_t no_gumbel_noise;      /* 1: MCTS_Gumbel(use_gumbel_noise=False), the class default (MCTS_Gumbel.py:157,592-596): no Gumbel
                                     variates are added to the root logits and no RNG event is consumed.  Self_Play passes True (:64) */
    uint32_t first_game_seq;      /* game sequence number of the first game of every slot (RNG streams are keyed by (seed, slot,
                                     game_seq)): a resumed generation passes the games already in the replay file so that no game
                                     is replayed (Self_Play.py:267-272 resumes by count; its workers reseed from OS entropy, :221) */
    int64_t games_budget;         /* continuous self-play only: > 0 = play exactly this many games — slot g plays its k-th game
                                     (k = game_seq - first_game_seq) iff k * n_games + g < games_budget, then halts — so a generation
                                     is the FIRST games_budget games STARTED, all run to completion (Self_Play.py:346-408), not the
                                     first ones to finish; 0 = slots restart forever */
    double tau;                   /* MCTS(tau=...) (MCTS.py:116-120,602-610): < 0 = Self_Play's schedule (tau 1 for the first
                                     num_explore_actions plies of each player, then 0); 0 = most visited move; > 0 = sample with
                                     weights N^(1/tau).  gaz_engine_set_hyperparams changes it between runs */
    double move_time_limit;       /* train_config["MCTS_time_limit"] in seconds (Self_Play.py:35,100-112), 0 = none.  PUCT: every game's move ends when its
                                     own wall clock since the move began passes the limit or run_iterations are used up, whichever comes first — at
                                     least one simulation (MCTS.py:559-560); results then depend on timing, as in the reference.  Gumbel: any
                                     limit makes every move run 3 x its legal moves iterations ("Time limit isn't allowed for gumbel",
                                     MCTS_Gumbel.py:576-578).  Continuous self-play only */
    int32_t game_groups;          /* scheduling only — no game depends on it.  K >= 2: the games run as K groups of consecutive slots, each with its own
                                     stream, evaluator batch and launch per wave, stepped alternately: the trunk tiles of one group fill the chip while
                                     another group's tree step starts or its heads run (continuous self-play with a built-in evaluator only; the
                                     wave_begin / batch API is refused).  1 = one batch.  0 = automatic: 2 where measured to pay (Connect4 PUCT +
                                     ResNet from 3072 games: +9.7 % evaluations/s; Gomoku PUCT + ResNet from 2048 games: +8.6 %; Connect4 Gumbel + ResNet from 6144 games: +6 %), else 1.  With the evaluation cache every group keeps a table of its own */
} gaz_engine_config;

/* MCTS.update_hyperparams(**kwargs) (MCTS.py:134-168) / MCTS_Gumbel.update_hyperparams (MCTS_Gumbel.py:186-210): values take
 * effect at the next launch.  A NaN double / negative int32 field means "unchanged" (kwargs.get(...) is None). */
typedef struct {
    uint32_t struct_size;         /* = sizeof(gaz_search_hyperparams) */
    int32_t use_dirichlet;        /* < 0 unchanged */
    double c_puct_init, c_puct_base, dirichlet_alpha, dirichlet_epsilon;
    double tau;                   /* as gaz_engine_config.tau; NaN unchanged */
    int32_t gumbel_m;             /* < 0 unchanged */
    int32_t run_iterations;       /* <= 0 unchanged */
    double c_visit, c_scale;
} gaz_search_hyperparams;

typedef struct {
    const char* name;             /* e.g. "stem.conv.weight" — see grok_alpha_zero_amd/net.py */
    const float* data;            /* host pointer, float32, C-contiguous */
    int64_t numel;
} gaz_tensor;

/* layout of one finished-game record in the byte blob returned by drain_finished */
typedef struct {
    int32_t record_bytes, max_T, A, t_pad;
    int32_t off_hdr, off_actions, off_q, off_root_visits, off_evals, off_policy, off_N, off_W, off_P;
} gaz_record_layout;

int gaz_engine_abi_version(void);                   /* GAZ_ENGINE_ABI_VERSION of the library */
int gaz_engine_config_size(void);                   /* sizeof(gaz_engine_config) of the library */
int gaz_engine_create(const gaz_engine_config* cfg, gaz_engine** out);
void gaz_engine_destroy(gaz_engine* h);
const char* gaz_engine_last_error(gaz_engine* h);   /* h may be NULL: last create() error */

int gaz_engine_load_weights(gaz_engine* h, const gaz_tensor* tensors, int32_t n);
int gaz_engine_reset_games(gaz_engine* h, const int32_t* slots, int32_t n);   /* slots NULL = all */

/* synchronous per-move API (cfg.sync_moves = 1) */
int gaz_engine_run_move(gaz_engine* h, int32_t* n_waiting);                    /* runs waves until every live game finished its MCTS.run */
int gaz_engine_get_root_stats(gaz_engine* h, uint32_t* out_N, float* out_W, float* out_P, float* out_policy,
                              uint32_t* out_root_visits, float* out_q, int32_t* out_chosen, int32_t* out_phase);
                              /* [n_games][A] x4, [n_games] x4; any pointer may be NULL */
int gaz_engine_apply_moves(gaz_engine* h, const int32_t* moves);               /* moves NULL / entry < 0 = play the sampled move */

/* continuous device-resident self-play (cfg.sync_moves = 0) */
int gaz_engine_run_waves(gaz_engine* h, int32_t n_waves);

/* external evaluator: wave_begin leaves the batch in HBM, wave_end consumes policy/value written by the caller */
int gaz_engine_wave_begin(gaz_engine* h);
int gaz_engine_wave_end(gaz_engine* h);
int gaz_engine_batch_ptrs(gaz_engine* h, void** d_inputs_i8, void** d_policy_f32, void** d_value_f32);   /* device pointers */
int gaz_engine_read_batch(gaz_engine* h, int8_t* inputs, int32_t* pending);    /* host copies: [n_games][H*W*C], [n_games] */
int gaz_engine_write_outputs(gaz_engine* h, const float* policy, const float* value);   /* host -> device rows */

/* place one slot at the position reached by `n` actions from the empty board (new roots are built there) */
int gaz_engine_set_position(gaz_engine* h, int32_t slot, const int32_t* actions, int32_t n);
/* iteration_limit of the following MCTS.run calls (<= 0: unchanged); tau_mode -1 = Self_Play schedule, 0 / 1 = fixed tau */
int gaz_engine_set_search_params(gaz_engine* h, int32_t run_iterations, int32_t tau_mode);
/* MCTS.update_hyperparams / MCTS_Gumbel.update_hyperparams for every tree of the engine (see gaz_search_hyperparams) */
int gaz_engine_set_hyperparams(gaz_engine* h, const gaz_search_hyperparams* hp);
/* MCTS.run(time_limit=...) (MCTS.py:528-563): stop != 0 makes every running search finish its move at the next launch, as the
 * reference's `time.time() - start_time < time_limit` test does between iterations; stop = 0 re-arms.  The host owns the clock. */
int gaz_engine_stop_search(gaz_engine* h, int32_t stop);

/* sync + single_tree engines idle after set_position / apply_moves; this starts MCTS.run for the idle slots */
int gaz_engine_start_search(gaz_engine* h);

/* Game-rules probe: n_positions positions, each given as n_actions[p] action indices (row p of actions[n_positions][stride]) played
 * from the empty board, first mover = -1.  The DEVICE rule code the search uses answers, per position: board int8 [H*W]; legal
 * uint8 [A] mask (get_legal_actions_MCTS); winner = check_win_MCTS after the last action (-2 running, -1 / 1 winner, 0 draw; -99 =
 * the history was not legal); input int8 [H*W*C] (get_input_state_MCTS); terminal int32 [A]: -1 not terminal, 1 the move wins, 0 it
 * draws (get_terminal_actions_fn, MCTS.py:247-294; all -1 for a finished position); and, when policy_in f32 [n][A] is given,
 * policy_out f32 [n][A] = get_legal_actions_policy_MCTS(..., normalize=True): policy at the legal actions / their sum, 0 elsewhere.
 * Output pointers may be NULL.  Reference: the static *_MCTS methods of the Game plugin (Guide.py:135-283), Game_Tester.py:297-405. */
int gaz_engine_probe_rules(gaz_engine* h, const int32_t* actions, const int32_t* n_actions, int32_t n_positions, int32_t stride,
                           int8_t* board, uint8_t* legal, int32_t* winner, int8_t* input, int32_t* terminal,
                           const float* policy_in, float* policy_out);

/* run the built-in evaluator on a host batch: inputs int8 [n][H*W*C] -> policy f32 [n][A], value f32 [n]; n <= n_games */
int gaz_engine_evaluate(gaz_engine* h, const int8_t* inputs, int32_t n, float* policy, float* value, int32_t repeats, double* ms_per_batch);

/* diagnostics for the numerics tests: the flat head features the last gaz_engine_evaluate left in HBM — for the Connect4 network
 * relu(bn0(conv3x3(x) + b)) of the policy and the value head, f32 [n][H*W*8] each (Connect4/Build_Model.py:41-47,62-66), i.e. the
 * output of stem + every residual block + the heads' first convolution.  *_row_floats receive the row length; p / v may be NULL. */
int gaz_engine_read_head_features(gaz_engine* h, int32_t n, float* p_feat, float* v_feat, int32_t* p_row_floats, int32_t* v_row_floats);

int gaz_engine_record_layout(gaz_engine* h, gaz_record_layout* out);
int gaz_engine_drain_finished(gaz_engine* h, void* out, int32_t max_records, int32_t* n_out);
int gaz_engine_get_stats(gaz_engine* h, uint64_t out[16]);  /* [0..5] game_stats, [6] evaluator calls, [7] simulations,
                                                               [8] plies played (= positions, incl. games in progress), [9] waves launched,
                                                               [10] evaluations answered by the evaluation cache, [11] groups of the group pipeline (0 = off),
                                                               [12] 1 = tree step and trunk kernel run as ONE fused launch,
                                                               [13] trunk workgroups of fused launches that gave up waiting for their games (see
                                                                    gaz_engine_debug_fused_fault); non-zero = the engine has fallen back to separate launches ([12] says whether it still is: the
                                                                    one-launch form is tried again after 20000 waves, at most twice),
                                                               [14] game groups (gaz_engine_config::game_groups as resolved; 0 = one batch): with groups, [0..8], [10], [13]
                                                                    are sums over the groups and [12] says that every group runs the one-launch form */
int gaz_engine_synchronize(gaz_engine* h);

/* Connect4 PUCT with the ResNet evaluator runs the tree step and the trunk kernel of a wave as ONE launch (k_wave_trunk: the trunk
 * starts on the boards whose games are done while the slow games still search); on = 0 launches them separately (same results bit
 * for bit; bench.py uses it to time the trunk kernel on its own).  Scheduling only. */
int gaz_engine_set_fused_wave(gaz_engine* h, int32_t on);

/* The fused launch hands leaf rows from tree blocks to trunk workgroups INSIDE one running kernel, which assumes that the tree blocks (lowest
 * block indices) become resident before the trunk workgroups that wait for them — HIP promises no dispatch order.  The wait is therefore
 * bounded (20 ms): a trunk workgroup that runs out of time leaves its boards unevaluated and marks them, the games keep their requests
 * pending and are evaluated by the next wave (no result changes), and at its next synchronisation point (synchronize, get_stats,
 * drain_finished) the engine switches to separate launches (get_stats [13] counts the workgroups that gave up) — for 20000 waves, then the one-launch
 * form is tried again; after the third give-up for good.  This hook makes every trunk workgroup with index % mod == 1 behave
 * as if its wait had timed out (mod = 0: off), so that the recovery path can be tested where the assumption holds. */
int gaz_engine_debug_fused_fault(gaz_engine* h, int32_t mod);

/* Continuous self-play with a games_budget: towards the end of a generation more and more slots have played their last game, but a
 * wave still steps and evaluates every slot.  repack moves the games that still run into the lowest slots (tree arena slice, records,
 * pending evaluator rows; a game keeps its identity) and shrinks all later launches to them.  *n_active = games still running,
 * *n_launch = slots the launches cover from now on.  Results do not change.  (The reference's counterpart is simply that finished
 * worker processes stop asking the inference server, Self_Play.py:380-400.)
 * After a repack a PHYSICAL slot index no longer identifies a game (records carry the game's own slot id): gaz_engine_set_position and
 * gaz_engine_reset_games with a slot list are refused from then on; gaz_engine_reset_games(h, NULL, 0) restarts every slot and makes the
 * launches cover all of them again.  The batch-level calls (read_batch / write_outputs / get_root_stats) keep addressing physical rows. */
int gaz_engine_repack(gaz_engine* h, int32_t* n_active, int32_t* n_launch);

/* The action history of every slot's game in progress — game.action_history of the reference's Game objects, as action indices; what
 * gaz_engine_set_position takes.  n_hist int32 [n_games] (0 for a halted slot), hist uint8 [n_games][stride] with stride >= the game's
 * max_T (gaz_record_layout.max_T).  Synchronises the engine's stream.  bench.py draws its staggered start from it. */
int gaz_engine_read_positions(gaz_engine* h, int32_t* n_hist, uint8_t* hist, int32_t stride);

/* measurement hooks (bench.py): HIP-event timing of the kernels launched on the engine's stream */
int gaz_engine_timing_reset(gaz_engine* h, int32_t enable);
/* the kernel priced against the roofline: its name (copied into `name`) and the algorithmic FLOPs of one launch */
int gaz_engine_dominant_kernel(gaz_engine* h, char* name, int32_t cap, double* flops_per_launch);
/* sums over the TIMED waves (run_waves brackets every 8th wave: an event record costs a barrier packet): tree-kernel ms, evaluator
 * ms, ms and launch count of the dominant kernel, number of timed waves.  With game groups (gaz_engine_config::game_groups) the ms are sums
 * over the groups' launches — which overlap in time, so they are kernel time, not wall clock — and gaz_engine_dominant_kernel prices ONE group's
 * launch; a per-kernel roofline is measured on an engine with game_groups = 1 (bench.py does) */
int gaz_engine_timing_get(gaz_engine* h, double* ms_tree, double* ms_eval, double* ms_dominant, int64_t* n_dominant, int64_t* n_waves);

#ifdef __cplusplus
}
#endif
#endif
