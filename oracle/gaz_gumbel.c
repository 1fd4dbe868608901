/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gaz_det.h header).
 *
 * gaz_gumbel.c — CPU restatement of class MCTS_Gumbel, /root/reference/MCTS_Gumbel.py:19-733, and of the
 * use_gumbel branch of Self_Play.play (Self_Play.py:58-69,108-112,151-153).  Arithmetic types follow what the
 * reference computes under numpy without Numba (see gaz_puct.c header): float32 statistics, float64 softmax.
 * exp/log: gaz_exp / gaz_log (det math) by default, libm with use_libm — both are checked against the fixtures.
 * np.argsort ties: stable ascending (documented tie rule, gaz_puct.c header).
 */
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include "gaz_puct.h"
#include "gaz_selfplay.h"

#define MAXA 225
#define F32_EPS 1.1920928955078125e-07f

static int g_libm = 0;
static double xexp(double x) { return g_libm ? exp(x) : gaz_exp(x); }

typedef struct gnode {
    struct gnode* parent; int child_id;
    int8_t* board; int* history; int n_history; int current_player;
    int n;                       /* len(child_visits) */
    struct gnode** children;     /* [n], NULL = not expanded (python None) */
    int* actions;                /* child_legal_actions dict: index -> action */
    uint32_t* visits; float* values; float* raw_values; float* logits;
    int is_terminal;
} gnode;

typedef struct {
    gaz_game_desc g;
    const int8_t* game_board; const int* game_history; const int* game_n_history; const int* game_next_player;
    gaz_eval_fn eval; void* eval_ctx;
    int m; double c_visit, c_scale; int use_gumbel_noise, use_softmax;
    gnode* root; uint64_t root_visits;
    gaz_event* ev;               /* stream shared by the per-move trees of one game */
    uint64_t n_evals;
} ggumbel;

static void* xc(size_t n, size_t s) { void* p = calloc(n ? n : 1, s); if (!p) abort(); return p; }

static gnode* gnode_new(ggumbel* t, int child_id, const int8_t* board, const int* hist, int n_hist, int extra, int player,
                        const int* actions, int n, const float* logits, int is_terminal, gnode* parent) {
    gnode* x = (gnode*)xc(1, sizeof(gnode));
    int HW = t->g.H * t->g.W;
    x->child_id = child_id; x->parent = parent; x->current_player = player; x->is_terminal = is_terminal; x->n = n;
    if (board) { x->board = (int8_t*)xc(HW, 1); memcpy(x->board, board, HW); }
    x->n_history = n_hist + (extra >= 0);
    x->history = (int*)xc(x->n_history + 1, sizeof(int));
    if (n_hist) memcpy(x->history, hist, sizeof(int) * n_hist);
    if (extra >= 0) x->history[n_hist] = extra;
    x->children = (gnode**)xc(n, sizeof(gnode*)); x->actions = (int*)xc(n, sizeof(int));
    x->visits = (uint32_t*)xc(n, 4); x->values = (float*)xc(n, 4); x->raw_values = (float*)xc(n, 4); x->logits = (float*)xc(n, 4);
    if (n && actions) memcpy(x->actions, actions, sizeof(int) * n);
    if (n && logits) memcpy(x->logits, logits, 4 * n);
    return x;
}
static void gnode_free(gnode* x) {
    if (!x) return;
    for (int i = 0; i < x->n; ++i) gnode_free(x->children[i]);
    free(x->board); free(x->history); free(x->children); free(x->actions); free(x->visits); free(x->values);
    free(x->raw_values); free(x->logits); free(x);
}

/* get_terminal_actions_fn (MCTS_Gumbel.py:281-318): order of appearance, NOT sorted */
static int g_terminal_actions(ggumbel* t, const int8_t* board, const int* legal, int n_legal, int next_player, int* acts, float* mask) {
    int HW = t->g.H * t->g.W, n = 0; int8_t tmp[MAXA];
    for (int i = 0; i < n_legal; ++i) {
        memcpy(tmp, board, HW); gaz_do_action(&t->g, tmp, legal[i], next_player);
        int r = gaz_check_win(&t->g, tmp, next_player, legal[i]);
        if (r == GAZ_RUNNING) continue;
        acts[n] = legal[i]; mask[n++] = (r == next_player) ? 1.0f : 0.0f;
    }
    return n;
}

/* _back_propagate (MCTS_Gumbel.py:530-546) */
static void g_backprop(ggumbel* t, gnode* node, float value, uint32_t visits) {
    while (node->parent) {
        int id = node->child_id; node = node->parent;
        node->values[id] = node->values[id] + value; node->visits[id] += visits; value = value * -1.0f;
    }
    t->root_visits += visits;
}

/* softmax (MCTS_Gumbel.py:82-88), float64 */
static void g_softmax(const double* x, int n, double* out) {
    double mx = x[0]; for (int i = 1; i < n; ++i) if (x[i] > mx) mx = x[i];
    double c = -mx;
    for (int i = 0; i < n; ++i) out[i] = xexp(x[i] + c);
    double s = gaz_np_sum_f64(out, n);
    for (int i = 0; i < n; ++i) out[i] = out[i] / s;
}

/* compute_pi(use_softmax=True) with compute_v_mix / rescale_q / sigma (MCTS_Gumbel.py:91-148) */
static void g_compute_pi(const float* raw_values, const float* q, const float* logits, const uint32_t* visits, int n,
                         uint32_t N_b, double c_visit, double c_scale, float* pi) {
    double l64[MAXA], p64[MAXA]; float probs[MAXA], tmp[MAXA], cq[MAXA];
    for (int i = 0; i < n; ++i) l64[i] = (double)logits[i];
    g_softmax(l64, n, p64);
    for (int i = 0; i < n; ++i) probs[i] = (float)p64[i];
    /* compute_v_mix */
    uint64_t sum_visits = 0; for (int i = 0; i < n; ++i) sum_visits += visits[i];
    for (int i = 0; i < n; ++i) tmp[i] = visits[i] > 0 ? probs[i] : 0.0f;
    float sum_probs = gaz_np_sum_f32(tmp, n);
    for (int i = 0; i < n; ++i) tmp[i] = visits[i] > 0 ? (probs[i] * q[i]) / sum_probs : 0.0f;
    float weighted_q = gaz_np_sum_f32(tmp, n);
    double wq = (double)weighted_q * (double)sum_visits;
    for (int i = 0; i < n; ++i) {
        float vmix = (float)(((double)raw_values[i] + wq) / (double)(sum_visits + 1));
        cq[i] = visits[i] > 0 ? q[i] : vmix;
    }
    /* rescale_q */
    float mn = cq[0], mx = cq[0];
    for (int i = 1; i < n; ++i) { if (cq[i] < mn) mn = cq[i]; if (cq[i] > mx) mx = cq[i]; }
    float den = (mx - mn) > F32_EPS ? (mx - mn) : F32_EPS;
    double sg = (c_visit + (double)N_b) * c_scale;               /* sigma: (c_visit + N_b) * c_scale * q, float64 */
    for (int i = 0; i < n; ++i) { float r = (cq[i] - mn) / den; l64[i] = (double)logits[i] + sg * (double)r; }
    g_softmax(l64, n, p64);
    for (int i = 0; i < n; ++i) pi[i] = (float)p64[i];
}

/* stablemax (MCTS_Gumbel.py:77-80; Net/Stablemax.py without the eps): s(x) = x + 1 (x >= 0) else 1 / (1 - x + eps32), normalised */
static void g_stablemax_f32(const float* x, int n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = x[i] >= 0.0f ? x[i] + 1.0f : 1.0f / ((1.0f - x[i]) + F32_EPS);
    float s = gaz_np_sum_f32(out, n);
    for (int i = 0; i < n; ++i) out[i] = out[i] / s;
}
static void g_stablemax_f64(const double* x, int n, double* out) {
    for (int i = 0; i < n; ++i) out[i] = x[i] >= 0.0 ? x[i] + 1.0 : 1.0 / ((1.0 - x[i]) + (double)F32_EPS);
    double s = gaz_np_sum_f64(out, n);
    for (int i = 0; i < n; ++i) out[i] = out[i] / s;
}

/* compute_pi(use_softmax=False) (MCTS_Gumbel.py:144-148) as numpy 2 evaluates it without Numba: probs = stablemax(float32 logits);
 * sigma(...) is float64 because (c_visit + N_b) is a NumPy float64 scalar, so logits + sigma and the second stablemax are float64
 * and the result is NOT cast back to float32 */
static void g_compute_pi_stable(const float* raw_values, const float* q, const float* logits, const uint32_t* visits, int n,
                                uint32_t N_b, double c_visit, double c_scale, double* pi) {
    double l64[MAXA]; float probs[MAXA], tmp[MAXA], cq[MAXA];
    g_stablemax_f32(logits, n, probs);
    uint64_t sum_visits = 0; for (int i = 0; i < n; ++i) sum_visits += visits[i];
    for (int i = 0; i < n; ++i) tmp[i] = visits[i] > 0 ? probs[i] : 0.0f;
    float sum_probs = gaz_np_sum_f32(tmp, n);
    for (int i = 0; i < n; ++i) tmp[i] = visits[i] > 0 ? (probs[i] * q[i]) / sum_probs : 0.0f;
    float weighted_q = gaz_np_sum_f32(tmp, n);
    double wq = (double)weighted_q * (double)sum_visits;
    for (int i = 0; i < n; ++i) {
        float vmix = (float)(((double)raw_values[i] + wq) / (double)(sum_visits + 1));
        cq[i] = visits[i] > 0 ? q[i] : vmix;
    }
    float mn = cq[0], mx = cq[0];
    for (int i = 1; i < n; ++i) { if (cq[i] < mn) mn = cq[i]; if (cq[i] > mx) mx = cq[i]; }
    float den = (mx - mn) > F32_EPS ? (mx - mn) : F32_EPS;
    double sg = (c_visit + (double)N_b) * c_scale;
    for (int i = 0; i < n; ++i) { float r = (cq[i] - mn) / den; l64[i] = (double)logits[i] + sg * (double)r; }
    g_stablemax_f64(l64, n, pi);
}

static void g_mean_q(const gnode* x, float* q) {   /* mean_values + q_transform (MCTS_Gumbel.py:238-240, 91-97) */
    for (int i = 0; i < x->n; ++i) {
        float mean = x->visits[i] > 0 ? (float)((double)x->values[i] / (double)x->visits[i]) : -1.0f;
        q[i] = (mean - (-1.0f)) / 2.0f;
    }
}

/* deterministic_selection (MCTS_Gumbel.py:228-243) */
static int g_det_select(ggumbel* t, const gnode* x) {
    float q[MAXA], pi[MAXA]; uint32_t nb = 0; uint64_t sv = 0;
    g_mean_q(x, q);
    for (int i = 0; i < x->n; ++i) { if (x->visits[i] > nb) nb = x->visits[i]; sv += x->visits[i]; }
    double pi64[MAXA];
    if (t->use_softmax) { g_compute_pi(x->raw_values, q, x->logits, x->visits, x->n, nb, t->c_visit, t->c_scale, pi); for (int i = 0; i < x->n; ++i) pi64[i] = (double)pi[i]; }
    else g_compute_pi_stable(x->raw_values, q, x->logits, x->visits, x->n, nb, t->c_visit, t->c_scale, pi64);
    int best = 0; double bs = 0;
    for (int i = 0; i < x->n; ++i) {
        double s = pi64[i] - (double)x->visits[i] / (double)(1 + sv);
        if (i == 0 || s > bs) { bs = s; best = i; }
    }
    return best;
}

static gnode* g_expand(ggumbel* t, gnode* node, int index, float* value, uint32_t* visits);

/* create_expand_root (MCTS_Gumbel.py:320-389) */
static void g_create_root(ggumbel* t) {
    int HW = t->g.H * t->g.W, n_hist = *t->game_n_history, next_player = *t->game_next_player;
    int legal[MAXA]; int n_legal = gaz_legal_actions(&t->g, t->game_board, legal);
    int tacts[MAXA]; float tmask[MAXA];
    int nt = g_terminal_actions(t, t->game_board, legal, n_legal, next_player, tacts, tmask);
    t->root_visits = 0;
    if (nt > 0) {
        int any = 0; for (int i = 0; i < nt; ++i) if (tmask[i] == 1.0f) any = 1;
        float pol[MAXA]; for (int i = 0; i < nt; ++i) pol[i] = any ? tmask[i] / (float)nt : 1.0f / (float)nt;
        t->root = gnode_new(t, 0, NULL, t->game_history, n_hist, -1, -next_player, tacts, nt, pol, GAZ_NOT_TERMINAL, NULL);
        memcpy(t->root->raw_values, tmask, 4 * nt);
        for (int i = 0; i < nt; ++i) {
            gnode* c = gnode_new(t, i, NULL, t->game_history, n_hist, tacts[i], -next_player, NULL, 0, NULL,
                                 tmask[i] == 1.0f ? next_player : 0, t->root);
            t->root->children[i] = c;
            g_backprop(t, c, any ? 1.0f : 0.0f, 1);
        }
    } else {
        int8_t state[MAXA * 4]; float policy[MAXA], value, lg[MAXA];
        gaz_input_state(&t->g, t->game_board, -next_player, t->game_history, n_hist, state);
        t->eval(t->eval_ctx, state, HW * t->g.C, policy, &value); t->n_evals++;
        for (int i = 0; i < n_legal; ++i) lg[i] = policy[legal[i]];              /* normalize=False: raw logits */
        t->root = gnode_new(t, 0, t->game_board, t->game_history, n_hist, -1, -next_player, legal, n_legal, lg, GAZ_NOT_TERMINAL, NULL);
    }
}

/* _expand / _expand_with_terminal_actions (MCTS_Gumbel.py:391-528) */
static gnode* g_expand(ggumbel* t, gnode* node, int index, float* value, uint32_t* visits) {
    int HW = t->g.H * t->g.W;
    int action = node->actions[index];
    int8_t cb[MAXA]; memcpy(cb, node->board, HW);
    gaz_do_action(&t->g, cb, action, -node->current_player);
    int legal[MAXA]; int n_legal = gaz_legal_actions(&t->g, cb, legal);
    int tacts[MAXA]; float tmask[MAXA];
    int nt = g_terminal_actions(t, cb, legal, n_legal, node->current_player, tacts, tmask);
    if (nt > 0) {
        int any = 0; for (int i = 0; i < nt; ++i) if (tmask[i] == 1.0f) any = 1;
        float pol[MAXA];
        for (int i = 0; i < nt; ++i) pol[i] = any ? tmask[i] / (float)nt : 1.0f / (float)nt;   /* len(terminal_mask == 1) quirk (:398) */
        gnode* tp = gnode_new(t, index, cb, node->history, node->n_history, action, -node->current_player, tacts, nt, pol,
                              GAZ_NOT_TERMINAL, node);
        node->children[index] = tp;
        memcpy(tp->raw_values, tmask, 4 * nt);                                    /* :427; visits / values stay 0 */
        for (int i = 0; i < nt; ++i)
            tp->children[i] = gnode_new(t, i, NULL, tp->history, tp->n_history, tacts[i], node->current_player, NULL, 0, NULL,
                                        tmask[i] == 1.0f ? node->current_player : 0, tp);
        *value = any ? -(float)nt : 0.0f; *visits = (uint32_t)nt;
        return tp;
    }
    int hist[512]; memcpy(hist, node->history, sizeof(int) * node->n_history); hist[node->n_history] = action;
    int8_t state[MAXA * 4]; float policy[MAXA], v, lg[MAXA];
    gaz_input_state(&t->g, cb, -node->current_player, hist, node->n_history + 1, state);
    t->eval(t->eval_ctx, state, HW * t->g.C, policy, &v); t->n_evals++;
    for (int i = 0; i < n_legal; ++i) lg[i] = policy[legal[i]];
    gnode* c = gnode_new(t, index, cb, node->history, node->n_history, action, -node->current_player, legal, n_legal, lg,
                         GAZ_NOT_TERMINAL, node);
    node->raw_values[index] = v;                                                  /* :516 */
    node->children[index] = c;
    *value = -v; *visits = 1;
    return c;
}

/* select (MCTS_Gumbel.py:245-260): returns the node to act on; *child_id set when it must be expanded */
static gnode* g_select(ggumbel* t, gnode* node, int* child_id) {
    for (;;) {
        int id = g_det_select(t, node);
        *child_id = id;
        if (!node->children[id]) return node;
        if (node->children[id]->is_terminal != GAZ_NOT_TERMINAL) return node->children[id];
        node = node->children[id];
    }
}

static void argsort_asc_f64(const double* v, int n, int* idx) {    /* stable ascending */
    for (int i = 0; i < n; ++i) idx[i] = i;
    for (int i = 1; i < n; ++i) { int k = idx[i], j = i - 1; while (j >= 0 && v[idx[j]] > v[k]) { idx[j + 1] = idx[j]; --j; } idx[j + 1] = k; }
}

/* run (MCTS_Gumbel.py:562-679).  rows in child order; returns the action of children[top_node_ids[0]] */
static int g_run(ggumbel* t, int iteration_limit, gaz_move_row* rows, int* n_rows) {
    gnode* r = t->root;
    int legal[MAXA]; int len_legal = gaz_legal_actions(&t->g, t->game_board, legal);
    int m = t->m; if (m > len_legal) m = len_legal;
    int n = r->n;
    float top_logits[MAXA]; int top_ids[MAXA]; float top_mean[MAXA]; int n_top = n;
    for (int i = 0; i < n; ++i) { top_logits[i] = r->logits[i]; top_ids[i] = i; top_mean[i] = r->values[i]; }
    if (t->use_gumbel_noise) {
        gaz_event e = *t->ev; e.purpose = GAZ_P_GUMBEL; t->ev->event++;
        for (int i = 0; i < n; ++i) top_logits[i] = (float)((double)top_logits[i] + gaz_gumbel(&e, (uint32_t)i));
    }
    int current_iteration = 0, phase = 0;
    while (len_legal > 1) {
        /* sequential_halving (:212-224) */
        double halved_m = (double)m / (double)(1 << phase); if (halved_m < 1.0) halved_m = 1.0;
        double score[MAXA]; int order[MAXA]; int take;
        uint32_t nb = 0; for (int i = 0; i < r->n; ++i) if (r->visits[i] > nb) nb = r->visits[i];
        if (phase == 0) { for (int i = 0; i < n_top; ++i) score[i] = (double)top_logits[i]; take = m; }
        else {
            double sg = (t->c_visit + (double)nb) * t->c_scale;
            for (int i = 0; i < n_top; ++i) { float qh = (top_mean[i] - (-1.0f)) / 2.0f; score[i] = (double)top_logits[i] + sg * (double)qh; }
            take = (int)halved_m;
        }
        argsort_asc_f64(score, n_top, order);
        if (take > n_top) take = n_top;
        double lg2;                                   /* np.log2(m): exact for powers of two, else log(m)/ln2 */
        if ((m & (m - 1)) == 0) { lg2 = 0.0; for (int mm = m; mm > 1; mm >>= 1) lg2 += 1.0; }
        else lg2 = g_libm ? log2((double)m) : gaz_log((double)m) / 0.6931471805599453;
        double denom = lg2 * halved_m;
        int vpc = (int)((double)iteration_limit / denom); if (vpc < 1) vpc = 1;
        float nl[MAXA]; int ni[MAXA];
        for (int i = 0; i < take; ++i) { nl[i] = top_logits[order[n_top - take + i]]; ni[i] = top_ids[order[n_top - take + i]]; }
        n_top = take; memcpy(top_logits, nl, 4 * n_top); memcpy(top_ids, ni, sizeof(int) * n_top);
        if (n_top == 1) break;
        if (n_top == 2 || n_top == 3) { vpc = (iteration_limit - current_iteration) / n_top; if (vpc < 1) vpc = 1; }
        for (int c = 0; c < n_top; ++c) {
            int id = top_ids[c]; float value; uint32_t visits;
            if (!r->children[id]) { gnode* nd = g_expand(t, r, id, &value, &visits); g_backprop(t, nd, value, visits); }
            for (int k = 0; k < vpc; ++k) {
                gnode* node = r->children[id]; int child_id = -1;
                if (node->is_terminal == GAZ_NOT_TERMINAL) node = g_select(t, node, &child_id);
                if (node->is_terminal != GAZ_NOT_TERMINAL) { value = (node->is_terminal == 1 || node->is_terminal == -1) ? 1.0f : 0.0f; visits = 1; }
                else node = g_expand(t, node, child_id, &value, &visits);
                g_backprop(t, node, value, visits);
                current_iteration++;
            }
        }
        for (int c = 0; c < n_top; ++c) top_mean[c] = (float)((double)r->values[top_ids[c]] / (double)r->visits[top_ids[c]]);
        phase++;
    }
    /* final policy (:653-675) */
    float mean[MAXA], q[MAXA], pi[MAXA]; uint32_t nb = 0;
    for (int i = 0; i < n; ++i) {
        mean[i] = r->visits[i] > 0 ? (float)((double)r->values[i] / (double)r->visits[i]) : -1.0f;
        q[i] = (mean[i] - (-1.0f)) / 2.0f;
        if (r->visits[i] > nb) nb = r->visits[i];
    }
    g_compute_pi(r->raw_values, q, r->logits, r->visits, n, nb, t->c_visit, t->c_scale, pi);
    for (int i = 0; i < n; ++i) {
        if (r->visits[i] == 0) mean[i] = pi[i];
        rows[i].action = r->actions[i]; rows[i].prob = (double)pi[i]; rows[i].winrate = (double)mean[i];
        rows[i].value = r->values[i]; rows[i].visits = r->visits[i]; rows[i].prior = r->logits[i];
        rows[i].root_visits = t->root_visits; rows[i].is_terminal = r->children[i] ? r->children[i]->is_terminal : GAZ_NOT_TERMINAL;
    }
    *n_rows = n;
    return r->actions[top_ids[0]];
}

/* Self_Play.play with use_gumbel = True (Self_Play.py:58-69, 108-153): a fresh MCTS_Gumbel every move */
int gaz_selfplay_game_gumbel(const gaz_sp_config* cfg, int m, double c_visit, double c_scale, int iteration_limit,
                             gaz_eval_fn eval, void* ctx, uint64_t seed, uint32_t slot, uint32_t game_seq, int use_libm,
                             gaz_sp_record* rec) {
    gaz_game_desc g = gaz_game(cfg->game_id);
    int HW = g.H * g.W, SZ = HW * g.C, A = g.A;
    int8_t board[225]; int history[256]; int n_history = 0, next_player = -1;
    memset(board, 0, sizeof(board));
    g_libm = use_libm & 1;                           /* bit 1 of use_libm: activation_fn = "stablemax" (Self_Play.py:69) */
    gaz_event ev; ev.key[0] = (uint32_t)seed; ev.key[1] = (uint32_t)(seed >> 32); ev.slot = slot; ev.game_seq = game_seq;
    ev.event = 0; ev.tree = 0; ev.purpose = 0;
    ggumbel t; memset(&t, 0, sizeof(t));
    t.g = g; t.game_board = board; t.game_history = history; t.game_n_history = &n_history; t.game_next_player = &next_player;
    t.eval = eval; t.eval_ctx = ctx; t.m = m; t.c_visit = c_visit; t.c_scale = c_scale; t.use_gumbel_noise = (use_libm & 4) ? 0 : 1;   /* bit 2: MCTS_Gumbel(use_gumbel_noise=False), MCTS_Gumbel.py:157,592 */
    t.use_softmax = (use_libm & 2) ? 0 : 1;
    t.ev = &ev;
    int winner = GAZ_RUNNING, T = 0, actions_count = 0; gaz_move_row rows[225]; int n_rows;
    g_create_root(&t);
    while (winner == GAZ_RUNNING && actions_count < cfg->max_actions) {
        gaz_input_state(&g, board, -next_player, history, n_history, rec->states + (size_t)T * SZ);
        uint64_t ev0 = t.n_evals;
        int action = g_run(&t, iteration_limit, rows, &n_rows);
        float* pol = rec->policies + (size_t)T * A; uint32_t* rn = rec->root_N + (size_t)T * A;
        float* rw = rec->root_W + (size_t)T * A; float* rp = rec->root_P + (size_t)T * A;
        for (int a = 0; a < A; ++a) { pol[a] = 0; rn[a] = 0; rw[a] = 0; rp[a] = 0; }
        float q = 0;
        for (int i = 0; i < n_rows; ++i) {
            pol[rows[i].action] = (float)rows[i].prob; rn[rows[i].action] = rows[i].visits; rw[rows[i].action] = rows[i].value;
            rp[rows[i].action] = rows[i].prior;
            if (rows[i].action == action) q = (float)rows[i].winrate;
        }
        rec->root_visits[T] = t.root_visits; rec->evals[T] = (uint32_t)(t.n_evals - ev0) + 0;
        rec->q[T] = q; rec->z[T] = (float)next_player;
        if (n_history == 0) action = gaz_opening_override(cfg, action, seed, slot, game_seq);
        rec->actions[T] = action; T++;
        gaz_do_action(&g, board, action, next_player); history[n_history++] = action; next_player = -next_player;
        winner = gaz_check_win(&g, board, -next_player, action);
        if (winner == GAZ_RUNNING) { gnode_free(t.root); t.root = NULL; t.m = m; g_create_root(&t); }   /* new MCTS_Gumbel (:152-153) */
        actions_count++;
        if (actions_count == cfg->max_actions) winner = 0;
    }
    if (winner == -1 && rec->z[T - 1] == -1.0f) { for (int i = 0; i < T; ++i) rec->z[i] *= -1.0f; }
    else if (winner == 0) { for (int i = 0; i < T; ++i) rec->z[i] = 0.0f; }
    for (int i = 0; i < T; ++i) rec->values[i] = 0.5f * (rec->z[i] + rec->q[i]);
    rec->T = T; rec->winner = winner; rec->total_evals = t.n_evals;
    gnode_free(t.root);
    g_libm = 0;
    return T;
}
