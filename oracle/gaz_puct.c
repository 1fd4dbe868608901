/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gaz_det.h header).
 *
 * gaz_puct.c — CPU restatement of class MCTS, /root/reference/MCTS.py:75-671.
 * Each function cites the lines it follows.  Arithmetic types follow what the
 * reference computes when run under numpy without Numba (the only way it can be
 * run in the build container; see DESIGN.md "Oracle"): PUCT scores in float64,
 * W accumulated in float32, N in uint32.
 *
 * Tie rule (documented divergence): np.argsort(x)[::-1] (MCTS.py:293,357,484) has
 * an implementation-defined order on ties (numpy SIMD sort vs Numba's insertion
 * sort).  The oracle, the fixture generator and the HIP engine all use
 * "stable ascending argsort, reversed": descending value, ties by HIGHER original
 * index first — what Numba's small-array insertion sort gives for MCTS.py:293.
 */
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include "gaz_puct.h"

#define MAXA 225

static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: OOM\n"); abort(); } return p; }
static void* xcalloc(size_t n, size_t s) { void* p = calloc(n ? n : 1, s); if (!p) { fprintf(stderr, "oracle: OOM\n"); abort(); } return p; }

/* Node.__init__ (MCTS.py:23-49) */
static gaz_node* node_new(gaz_puct* t, int child_id, const int8_t* board, const int* hist, int n_hist, int extra_action,
                          int current_player, const int* legal, int n_legal, const float* priors, int is_terminal,
                          gaz_node* parent) {
    gaz_node* n = (gaz_node*)xcalloc(1, sizeof(gaz_node));
    int HW = t->g.H * t->g.W;
    n->child_id = child_id;
    if (board) { n->board = (int8_t*)xmalloc(HW); memcpy(n->board, board, HW); }
    n->n_history = n_hist + (extra_action >= 0 ? 1 : 0);
    n->history = (int*)xmalloc(sizeof(int) * (n->n_history + 1));
    if (n_hist) memcpy(n->history, hist, sizeof(int) * n_hist);
    if (extra_action >= 0) n->history[n_hist] = extra_action;
    n->current_player = current_player;
    n->n_actions = n_legal;
    n->legal_actions = (int*)xmalloc(sizeof(int) * n_legal);
    if (n_legal) memcpy(n->legal_actions, legal, sizeof(int) * n_legal);
    n->children = (gaz_node**)xcalloc(n_legal, sizeof(gaz_node*));
    n->child_visits = (uint32_t*)xcalloc(n_legal, sizeof(uint32_t));
    n->child_values = (float*)xcalloc(n_legal, sizeof(float));
    n->child_prob_priors = (float*)xcalloc(n_legal, sizeof(float));
    if (priors && n_legal) memcpy(n->child_prob_priors, priors, sizeof(float) * n_legal);
    n->is_terminal = is_terminal;
    n->parent = parent;
    t->n_nodes++;
    return n;
}

static void node_free(gaz_node* n, gaz_node* keep) {
    if (!n || n == keep) return;
    for (int i = 0; i < n->n_children; ++i) node_free(n->children[i], keep);
    free(n->board); free(n->history); free(n->legal_actions); free(n->children);
    free(n->child_visits); free(n->child_values); free(n->child_prob_priors); free(n);
}

/* stable ascending argsort reversed (see file header) */
static void argsort_desc(const float* v, int n, int* idx) {
    for (int i = 0; i < n; ++i) idx[i] = i;
    for (int i = 1; i < n; ++i) { /* stable insertion sort ascending */
        int k = idx[i]; int j = i - 1;
        while (j >= 0 && v[idx[j]] > v[k]) { idx[j + 1] = idx[j]; --j; }
        idx[j + 1] = k;
    }
    for (int i = 0; i < n / 2; ++i) { int tmp = idx[i]; idx[i] = idx[n - 1 - i]; idx[n - 1 - i] = tmp; }
}

/* MCTS._get_best_PUCT_score_index (MCTS.py:172-191), numpy (non-fastmath) evaluation order:
 *   U = (P * (parent_visits**0.5 / (N + 1))) * (c_init + log((parent_visits + c_base + 1) / c_base))   [float64]
 *   Q = W (float32); Q[N>0] = float32(W / N) ; score = Q + U ; np.argmax -> first maximum.
 * use_libm = 1 uses libm sqrt/log exactly as CPython/numpy scalars do; 0 uses gaz_log (what the
 * HIP engine computes, bit for bit).  Both are checked against the fixtures. */
int gaz_puct_best_index(const float* priors, const float* values, const uint32_t* visits, int n,
                        uint64_t parent_visits, double c_init, double c_base, int use_libm) {
    double pv = (double)parent_visits;
    double s = use_libm ? pow(pv, 0.5) : gaz_sqrt(pv);
    double larg = (pv + c_base + 1.0) / c_base;
    double c = c_init + (use_libm ? log(larg) : gaz_log(larg));
    int best = 0; double best_score = 0.0;
    for (int i = 0; i < n; ++i) {
        double u = ((double)priors[i] * (s / (double)(visits[i] + 1u))) * c;
        float q = values[i];
        if (visits[i] > 0) q = (float)((double)values[i] / (double)visits[i]);
        double score = (double)q + u;
        if (i == 0 || score > best_score) { best = i; best_score = score; }
    }
    return best;
}

#ifndef GAZ_ORACLE_LIBM
#define GAZ_ORACLE_LIBM 0
#endif
static int g_use_libm = GAZ_ORACLE_LIBM;
void gaz_oracle_set_libm(int on) { g_use_libm = on; }

/* MCTS.get_terminal_actions_fn (MCTS.py:247-294).  Returns count; actions/mask sorted wins first. */
static int terminal_actions(gaz_puct* t, const int8_t* board, const int* legal, int n_legal, int next_player,
                            int* out_actions, float* out_mask) {
    int HW = t->g.H * t->g.W;
    int idx_a[MAXA]; float mask[MAXA]; int n = 0;
    int8_t tmp[MAXA];
    for (int i = 0; i < n_legal; ++i) {
        memcpy(tmp, board, HW);
        gaz_do_action(&t->g, tmp, legal[i], next_player);
        int result = gaz_check_win(&t->g, tmp, next_player, legal[i]);
        if (result == GAZ_RUNNING) continue;
        idx_a[n] = legal[i];
        if (result == next_player) { mask[n++] = 1.0f; if (t->fast_find_win) break; }
        else mask[n++] = 0.0f;
    }
    if (n == 0) return 0;
    int order[MAXA]; argsort_desc(mask, n, order);           /* MCTS.py:293-294 */
    for (int i = 0; i < n; ++i) { out_actions[i] = idx_a[order[i]]; out_mask[i] = mask[order[i]]; }
    return n;
}

/* MCTS._back_propagate (MCTS.py:513-526) */
static void back_propagate(gaz_puct* t, gaz_node* node, float value, uint32_t visits) {
    while (node->parent != NULL) {
        int id = node->child_id;
        node = node->parent;
        node->child_values[id] = node->child_values[id] + value;
        node->child_visits[id] += visits;
        value = value * -1.0f;
    }
    t->root_visits += visits;
}

/* policy -> sorted priors: get_legal_actions_policy_MCTS(normalize=True) (Connect4.py:280-299,
 * Gomoku.py:123-150, Tictactoe.py:191-209), MCTS._apply_dirichlet (MCTS.py:243-245),
 * argsort descending (MCTS.py:357-359, 484-487). */
static void make_priors(gaz_puct* t, const float* policy, const int* legal, int n_legal, int* sorted_actions, float* sorted_priors) {
    float p[MAXA];
    for (int i = 0; i < n_legal; ++i) p[i] = policy[gaz_policy_index(&t->g, legal[i])];
    float s = gaz_np_sum_f32(p, n_legal);
    for (int i = 0; i < n_legal; ++i) p[i] = p[i] / s;
    if (t->use_dirichlet) {
        double d[MAXA];
        gaz_event e = t->ev; e.purpose = GAZ_P_DIRICHLET; t->ev.event++;
        gaz_dirichlet(&e, t->dirichlet_alpha, n_legal, d);
        float one_minus = (float)(1.0 - t->dirichlet_epsilon);
        for (int i = 0; i < n_legal; ++i) {
            float a = one_minus * p[i];
            p[i] = (float)((double)a + t->dirichlet_epsilon * d[i]);
        }
    }
    int order[MAXA]; argsort_desc(p, n_legal, order);
    for (int i = 0; i < n_legal; ++i) { sorted_actions[i] = legal[order[i]]; sorted_priors[i] = p[order[i]]; }
}

/* MCTS.create_expand_root (MCTS.py:296-365) */
static void create_expand_root(gaz_puct* t) {
    node_free(t->root, NULL); t->root = NULL; t->root_visits = 0;
    int HW = t->g.H * t->g.W;
    int n_hist = *t->game_n_history;
    int next_player = *t->game_next_player;
    int legal[MAXA]; int n_legal = gaz_legal_actions(&t->g, t->game_board, legal);
    int tacts[MAXA]; float tmask[MAXA];
    int nt = terminal_actions(t, t->game_board, legal, n_legal, next_player, tacts, tmask);
    if (nt > 0) {
        int any_win = 0; for (int i = 0; i < nt; ++i) if (tmask[i] == 1.0f) any_win = 1;
        float value = any_win ? 1.0f : 0.0f;
        float pol[MAXA];
        for (int i = 0; i < nt; ++i) pol[i] = any_win ? tmask[i] / (float)nt : 1.0f / (float)nt;
        t->root = node_new(t, 0, NULL, t->game_history, n_hist, -1, -next_player, tacts, nt, pol, GAZ_NOT_TERMINAL, NULL);
        for (int i = 0; i < nt; ++i) {                              /* MCTS.py:331-344 */
            gaz_node* c = node_new(t, t->root->n_children, NULL, t->game_history, n_hist, tacts[i], -next_player,
                                   NULL, 0, NULL, tmask[i] == 1.0f ? next_player : 0, t->root);
            t->root->children[t->root->n_children++] = c;
            back_propagate(t, c, value, 1);
        }
    } else {
        int8_t state[MAXA * 4]; float policy[MAXA]; float value;
        gaz_input_state(&t->g, t->game_board, -next_player, t->game_history, n_hist, state);
        t->eval(t->eval_ctx, state, HW * t->g.C, policy, &value); t->n_evals++;
        int sa[MAXA]; float sp[MAXA];
        make_priors(t, policy, legal, n_legal, sa, sp);
        t->root = node_new(t, 0, t->game_board, t->game_history, n_hist, -1, -next_player, sa, n_legal, sp, GAZ_NOT_TERMINAL, NULL);
    }
}

/* MCTS._expand_with_terminal_actions (MCTS.py:367-428) */
static gaz_node* expand_with_terminal_actions(gaz_puct* t, gaz_node* node, const int8_t* board, int action,
                                              const int* tacts, const float* tmask, int nt, float* value, uint32_t* visits) {
    int any_win = 0; for (int i = 0; i < nt; ++i) if (tmask[i] == 1.0f) any_win = 1;
    float pol[MAXA]; int tp_value;
    if (any_win) {
        int k = nt;                       /* len(terminal_mask == 1) is the ARRAY length (MCTS.py:373) */
        tp_value = k; *visits = (uint32_t)k;
        for (int i = 0; i < nt; ++i) pol[i] = tmask[i] / (float)k;
    } else {
        tp_value = 0; *visits = (uint32_t)nt;
        for (int i = 0; i < nt; ++i) pol[i] = 1.0f / (float)nt;
    }
    gaz_node* tp = node_new(t, node->n_children, board, node->history, node->n_history, action, -node->current_player,
                            tacts, nt, pol, GAZ_NOT_TERMINAL, node);
    node->children[node->n_children++] = tp;
    for (int i = 0; i < nt; ++i) { tp->child_values[i] = tmask[i]; tp->child_visits[i] = 1; }   /* MCTS.py:398-401 */
    for (int i = 0; i < nt; ++i) {
        gaz_node* c = node_new(t, tp->n_children, NULL, tp->history, tp->n_history, tacts[i], node->current_player,
                               NULL, 0, NULL, tmask[i] == 1.0f ? node->current_player : 0, tp);
        tp->children[tp->n_children++] = c;
    }
    *value = (float)(-tp_value);
    return tp;
}

/* MCTS._expand (MCTS.py:434-511) */
static gaz_node* expand(gaz_puct* t, gaz_node* node, float* value, uint32_t* visits) {
    int HW = t->g.H * t->g.W;
    if (node->n_children >= node->n_actions || !node->board) { fprintf(stderr, "oracle: expand on exhausted node\n"); abort(); }
    int child_action = node->legal_actions[node->n_children];           /* popleft, MCTS.py:437 */
    int8_t child_board[MAXA];
    memcpy(child_board, node->board, HW);
    gaz_do_action(&t->g, child_board, child_action, -node->current_player);
    int legal[MAXA]; int n_legal = gaz_legal_actions(&t->g, child_board, legal);
    int tacts[MAXA]; float tmask[MAXA];
    int nt = terminal_actions(t, child_board, legal, n_legal, node->current_player, tacts, tmask);
    if (nt > 0) return expand_with_terminal_actions(t, node, child_board, child_action, tacts, tmask, nt, value, visits);

    int hist[512];
    memcpy(hist, node->history, sizeof(int) * node->n_history);
    hist[node->n_history] = child_action;
    int8_t state[MAXA * 4]; float policy[MAXA]; float v;
    gaz_input_state(&t->g, child_board, -node->current_player, hist, node->n_history + 1, state);
    t->eval(t->eval_ctx, state, HW * t->g.C, policy, &v); t->n_evals++;
    int sa[MAXA]; float sp[MAXA];
    make_priors(t, policy, legal, n_legal, sa, sp);
    gaz_node* child = node_new(t, node->n_children, child_board, node->history, node->n_history, child_action,
                               -node->current_player, sa, n_legal, sp, GAZ_NOT_TERMINAL, node);
    node->children[node->n_children++] = child;
    if (node->n_children == node->n_actions) { free(node->board); node->board = NULL; }   /* MCTS.py:501-509 */
    *value = -v; *visits = 1;
    return child;
}

/* MCTS._PUCT_select (MCTS.py:193-222) */
static gaz_node* puct_select(gaz_puct* t) {
    gaz_node* node = t->root;
    uint64_t parent_visits = t->root_visits;
    for (;;) {
        if (node->n_children > 0 && node->children[0]->is_terminal != GAZ_NOT_TERMINAL) {   /* terminal parent */
            gaz_node* cand[MAXA]; int nc = 0;
            if (gaz_np_sum_f32(node->child_values, node->n_actions) > 0.0f) {
                for (int i = 0; i < node->n_children; ++i) if (node->children[i]->is_terminal != 0) cand[nc++] = node->children[i];
            } else {
                for (int i = 0; i < node->n_children; ++i) cand[nc++] = node->children[i];
            }
            gaz_event e = t->ev; e.purpose = GAZ_P_TERMINAL_PICK; t->ev.event++;
            return cand[gaz_pick(&e, (uint32_t)nc)];                                        /* np.random.randint, MCTS.py:208 */
        }
        int best = gaz_puct_best_index(node->child_prob_priors, node->child_values, node->child_visits, node->n_actions,
                                       parent_visits, t->c_puct_init, t->c_puct_base, g_use_libm);
        if (best == node->n_children) return node;
        if (best > node->n_children) { fprintf(stderr, "oracle: PUCT picked an un-poppable child\n"); abort(); }
        parent_visits = node->child_visits[best];
        node = node->children[best];
    }
}

/* MCTS.run (MCTS.py:528-618) with time_limit=None */
int gaz_puct_run(gaz_puct* t, int iteration_limit, gaz_move_row* out_rows, int* n_rows) {
    int legal[MAXA]; int len_legal = gaz_legal_actions(&t->g, t->game_board, legal);
    if (len_legal == 1) iteration_limit = 1;                                   /* MCTS.py:543-546 */
    else if (iteration_limit < len_legal) iteration_limit = len_legal * 3;
    int fully_visited = 0;
    for (int it = 0; it < iteration_limit; ++it) {
        if (!fully_visited) {
            int has_zero = 0;
            for (int i = 0; i < t->root->n_actions; ++i) if (t->root->child_visits[i] == 0) has_zero = 1;
            if (!has_zero) fully_visited = 1;
        }
        gaz_node* node = fully_visited ? puct_select(t) : t->root;
        float value; uint32_t visits;
        if (node->is_terminal != GAZ_NOT_TERMINAL) {
            value = (node->is_terminal == 1 || node->is_terminal == -1) ? 1.0f : 0.0f; visits = 1;
        } else {
            node = expand(t, node, &value, &visits);
        }
        back_propagate(t, node, value, visits);
    }
    gaz_node* r = t->root;
    int n = r->n_children;
    if (n != r->n_actions) { fprintf(stderr, "oracle: root not fully expanded after run (%d/%d)\n", n, r->n_actions); abort(); }
    uint64_t sumv = 0; for (int i = 0; i < r->n_actions; ++i) sumv += r->child_visits[i];
    for (int i = 0; i < n; ++i) {                                              /* MCTS.py:591-600 */
        gaz_node* c = r->children[i];
        out_rows[i].action = c->history[c->n_history - 1];
        out_rows[i].prob = (double)r->child_visits[i] / (double)sumv;
        out_rows[i].winrate = (double)r->child_values[i] / (double)r->child_visits[i];
        out_rows[i].value = r->child_values[i];
        out_rows[i].visits = r->child_visits[i];
        out_rows[i].prior = r->child_prob_priors[i];
        out_rows[i].root_visits = t->root_visits;
        out_rows[i].is_terminal = c->is_terminal;
    }
    *n_rows = n;
    double w[MAXA];
    if (t->tau == 0.0) {                                                       /* MCTS.py:602-604 */
        int am = 0; for (int i = 1; i < n; ++i) if (r->child_visits[i] > r->child_visits[am]) am = i;
        for (int i = 0; i < n; ++i) w[i] = (i == am) ? 1.0 : 0.0;
    } else {                                                                   /* MCTS.py:606-610 */
        double ex = 1.0 / t->tau;
        /* np.float64 ** np.float64 is libm pow (use_libm); otherwise the device's exp(y log x) stand-in (gaz_pow) */
        double den = (ex == 1.0) ? (double)t->root_visits : (g_use_libm ? pow((double)t->root_visits, ex) : gaz_pow((double)t->root_visits, ex));
        for (int i = 0; i < n; ++i) {
            double num = (ex == 1.0) ? (double)r->child_visits[i] : (g_use_libm ? pow((double)r->child_visits[i], ex) : gaz_pow((double)r->child_visits[i], ex));
            w[i] = num / den;
        }
        double s = gaz_np_sum_f64(w, n);
        for (int i = 0; i < n; ++i) w[i] = w[i] / s;
    }
    /* np.random.choice(arange(n), size=1, replace=False, p=w) (MCTS.py:612): legacy path =
     * cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, side='right') */
    gaz_event e = t->ev; e.purpose = GAZ_P_MOVE; t->ev.event++;
    double u = gaz_uniform(&e);
    double cdf[MAXA]; double acc = 0.0;
    for (int i = 0; i < n; ++i) { acc = acc + w[i]; cdf[i] = acc; }
    double last = cdf[n - 1];
    int chosen = n - 1;
    for (int i = 0; i < n; ++i) { cdf[i] = cdf[i] / last; }
    for (int i = 0; i < n; ++i) if (cdf[i] > u) { chosen = i; break; }
    return out_rows[chosen].action;
}

/* MCTS._set_root + MCTS.prune_tree (MCTS.py:620-671).  The reference builds a new Root that
 * SHARES the child's children/value/visit arrays and leaves the old ancestors referenced through
 * child.parent (their stats keep being updated but are unreachable); re-rooting in place and
 * cutting the parent link is observationally identical. */
void gaz_puct_prune(gaz_puct* t, int action, int create_new_root) {
    if (!create_new_root) {
        gaz_node* r = t->root;
        for (int i = 0; i < r->n_children; ++i) {
            gaz_node* c = r->children[i];
            if (c->history[c->n_history - 1] == action) {
                uint64_t v = r->child_visits[c->child_id];                      /* MCTS.py:654 */
                node_free(r, c);
                c->parent = NULL; c->child_id = 0;
                if (c->is_terminal == GAZ_NOT_TERMINAL && c->n_children < c->n_actions) {
                    int HW = t->g.H * t->g.W;                                   /* board = game.board.copy(), MCTS.py:637 */
                    if (!c->board) c->board = (int8_t*)xmalloc(HW);
                    memcpy(c->board, t->game_board, HW);
                }
                t->root = c; t->root_visits = v;
                return;
            }
        }
    }
    create_expand_root(t);
}

gaz_puct* gaz_puct_create(int game_id, const int8_t* game_board, const int* game_history, const int* game_n_history,
                          const int* game_next_player, gaz_eval_fn eval, void* ctx,
                          double c_puct_init, double c_puct_base, int use_dirichlet, double alpha, double eps,
                          double tau, uint64_t seed, uint32_t slot, uint32_t game_seq, uint32_t tree) {
    gaz_puct* t = (gaz_puct*)xcalloc(1, sizeof(gaz_puct));
    t->g = gaz_game(game_id);
    t->game_board = game_board; t->game_history = game_history; t->game_n_history = game_n_history;
    t->game_next_player = game_next_player;
    t->eval = eval; t->eval_ctx = ctx;
    t->c_puct_init = c_puct_init; t->c_puct_base = c_puct_base;
    t->use_dirichlet = use_dirichlet; t->dirichlet_epsilon = eps;
    /* dirichlet_alpha * np.ones_like(legal_policy) is a float32 array (MCTS.py:244-245): alpha is rounded to f32 */
    t->dirichlet_alpha = (double)(float)alpha;
    gaz_puct_set_tau(t, tau);
    t->ev.key[0] = (uint32_t)seed; t->ev.key[1] = (uint32_t)(seed >> 32);
    t->ev.slot = slot; t->ev.game_seq = game_seq; t->ev.event = 0; t->ev.tree = tree; t->ev.purpose = 0;
    create_expand_root(t);                                                      /* MCTS.py:132 */
    return t;
}

/* MCTS.update_hyperparams(tau=...) (MCTS.py:163-168) / __init__ (MCTS.py:116-120) */
void gaz_puct_set_tau(gaz_puct* t, double tau) {
    if (tau != 0.0 && tau <= 5e-3) tau = 0.0;
    t->tau = tau;
}

void gaz_puct_destroy(gaz_puct* t) { if (!t) return; node_free(t->root, NULL); free(t); }

/* inspection helpers for tests */
uint64_t gaz_puct_root_visits(const gaz_puct* t) { return t->root_visits; }
uint64_t gaz_puct_n_evals(const gaz_puct* t) { return t->n_evals; }
uint32_t gaz_puct_event(const gaz_puct* t) { return t->ev.event; }
