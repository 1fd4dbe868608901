/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gaz_det.h header).
 *
 * gaz_selfplay.c — CPU restatement of Self_Play.__init__/play for the PUCT path,
 * /root/reference/Self_Play.py:16-57,71-175 (everything up to the HDF5 append), plus
 * the synthetic "hash evaluator" used as a bit-reproducible stand-in for session.run.
 */
#include <stdlib.h>
#include <stdio.h>
#include "gaz_puct.h"
#include "gaz_selfplay.h"

/* ---- hash evaluator: a deterministic function of the int8 input state -------------
 * policy[a] = ((h_a >> 8) + 1) * 2^-24 in (0, 1];  value = (h_v >> 8) * 2^-23 - 1 in [-1, 1)
 * h_x = fmix32(FNV-1a over the state bytes, seeded with salt ^ x * 0x9E3779B1).  Exactly
 * representable float32 outputs, so Python / C / HIP implementations agree bit for bit. */
static inline uint32_t hash_state(const int8_t* s, int n, uint32_t seed) {
    uint32_t h = 2166136261u ^ seed;
    for (int i = 0; i < n; ++i) { h ^= (uint8_t)s[i]; h *= 16777619u; }
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
void gaz_hash_eval(void* ctx, const int8_t* state, int n_state, float* policy, float* value) {
    const gaz_hash_eval_ctx* c = (const gaz_hash_eval_ctx*)ctx;
    for (int a = 0; a < c->A; ++a) {
        uint32_t h = hash_state(state, n_state, c->salt ^ ((uint32_t)(a + 1) * 0x9E3779B1u));
        policy[a] = (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
    }
    uint32_t hv = hash_state(state, n_state, c->salt ^ 0x51ED270Bu);
    *value = (float)(hv >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

/* Self_Play.py:130-140: at move 0, with opening_actions configured, the played action is drawn from
 * opening_actions (+ the search's own action with the remaining probability) by np.random.choice (game-level stream). */
int gaz_opening_override(const gaz_sp_config* cfg, int mcts_action, uint64_t seed, uint32_t slot, uint32_t game_seq) {
    if (cfg->n_opening <= 0) return mcts_action;
    int acts[9]; double w[9]; int n = cfg->n_opening; double sum = 0.0;
    for (int i = 0; i < n; ++i) { acts[i] = cfg->opening_actions[i]; w[i] = cfg->opening_weights[i]; sum = sum + w[i]; }
    if (sum < 1.0) { acts[n] = mcts_action; w[n] = 1.0 - sum; n++; }
    gaz_event e; e.key[0] = (uint32_t)seed; e.key[1] = (uint32_t)(seed >> 32); e.slot = slot; e.game_seq = game_seq;
    e.event = 0; e.tree = 2; e.purpose = GAZ_P_OPENING;
    double u = gaz_uniform(&e), cdf[9], acc = 0.0;
    for (int i = 0; i < n; ++i) { acc = acc + w[i]; cdf[i] = acc; }
    for (int i = 0; i < n; ++i) if (cdf[i] / cdf[n - 1] > u) return acts[i];
    return acts[n - 1];
}

/* Self_Play.play (Self_Play.py:71-175), use_gumbel = False. */
int gaz_selfplay_game(const gaz_sp_config* cfg, gaz_eval_fn eval, void* ctx, uint64_t seed, uint32_t slot,
                      uint32_t game_seq, gaz_sp_record* rec) {
    gaz_game_desc g = gaz_game(cfg->game_id);
    int HW = g.H * g.W, SZ = HW * g.C, A = g.A;
    int8_t board[225]; int history[256]; int n_history = 0; int next_player = -1;
    memset(board, 0, sizeof(board));

    /* Self_Play.__init__ (Self_Play.py:37-57): two trees on ONE game object; mcts1 first */
    gaz_puct* mcts[2];
    for (int k = 0; k < 2; ++k)
        mcts[k] = gaz_puct_create(cfg->game_id, board, history, &n_history, &next_player, eval, ctx,
                                  cfg->c_puct_init, cfg->c_puct_base, 1, cfg->dirichlet_alpha, 0.25, 1.0,
                                  seed, slot, game_seq, (uint32_t)k);
    int actions_count = 0, winner = GAZ_RUNNING, T = 0;
    gaz_move_row rows[225]; int n_rows;
    while (winner == GAZ_RUNNING && actions_count < cfg->max_actions) {
        if (T >= rec->cap_T) { fprintf(stderr, "oracle: record capacity\n"); abort(); }
        gaz_input_state(&g, board, -next_player, history, n_history, rec->states + (size_t)T * SZ);   /* :80 */
        int num = n_history;
        gaz_puct_set_tau(mcts[0], (num % 2 == 0 && num / 2 < cfg->num_explore_actions_first) ? 1.0 : 0.0);       /* :86-89 */
        gaz_puct_set_tau(mcts[1], ((num + 1) % 2 == 0 && (num + 1) / 2 < cfg->num_explore_actions_second) ? 1.0 : 0.0); /* :91-95 */
        gaz_puct* runner = (next_player == -1) ? mcts[0] : mcts[1];
        uint64_t ev0 = runner->n_evals;
        int action = gaz_puct_run(runner, cfg->run_iterations, rows, &n_rows);      /* :97-106 (caller passes int(1.5*limit)) */
        float* pol = rec->policies + (size_t)T * A;
        uint32_t* rn = rec->root_N + (size_t)T * A; float* rw = rec->root_W + (size_t)T * A; float* rp = rec->root_P + (size_t)T * A;
        for (int a = 0; a < A; ++a) { pol[a] = 0.0f; rn[a] = 0; rw[a] = 0.0f; rp[a] = 0.0f; }
        float q = 0.0f;
        for (int i = 0; i < n_rows; ++i) {
            pol[rows[i].action] = (float)rows[i].prob;                               /* compute_policy_improvement */
            rn[rows[i].action] = rows[i].visits; rw[rows[i].action] = rows[i].value; rp[rows[i].action] = rows[i].prior;
            if (rows[i].action == action) q = (float)rows[i].winrate;               /* :117-125 */
        }
        rec->root_visits[T] = rows[0].root_visits;
        rec->evals[T] = (uint32_t)(runner->n_evals - ev0);
        rec->q[T] = q; rec->z[T] = (float)next_player;                              /* :127 */
        if (n_history == 0) action = gaz_opening_override(cfg, action, seed, slot, game_seq);                       /* :130-140 */
        rec->actions[T] = action;                                                    /* the action PLAYED */
        T++;
        rec->T = T;                                                                  /* live progress (bench.py's CPU baseline reads it from another thread) */
        gaz_do_action(&g, board, action, next_player); history[n_history++] = action; next_player = -next_player;  /* :142 */
        winner = gaz_check_win(&g, board, -next_player, action);
        if (winner == GAZ_RUNNING) { gaz_puct_prune(mcts[0], action, cfg->create_new_root); gaz_puct_prune(mcts[1], action, cfg->create_new_root); }
        actions_count++;
        if (actions_count == cfg->max_actions) winner = 0;                           /* :156-157 */
    }
    if (winner == -1 && rec->z[T - 1] == -1.0f) { for (int i = 0; i < T; ++i) rec->z[i] *= -1.0f; }   /* :165-168 */
    else if (winner == 0) { for (int i = 0; i < T; ++i) rec->z[i] = 0.0f; }
    for (int i = 0; i < T; ++i) rec->values[i] = 0.5f * (rec->z[i] + rec->q[i]);   /* :172 */
    rec->T = T; rec->winner = winner;
    rec->total_evals = mcts[0]->n_evals + mcts[1]->n_evals;
    gaz_puct_destroy(mcts[0]); gaz_puct_destroy(mcts[1]);
    return T;
}
