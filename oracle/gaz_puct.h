/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gaz_det.h header).
 *
 * gaz_puct.h — CPU restatement of the reference's PUCT search, class MCTS in
 * /root/reference/MCTS.py:75-671, as a pointer tree that follows the Python
 * object graph one to one (Node MCTS.py:20-49, Root MCTS.py:52-72).  It is NOT
 * the device layout; the HIP engine is checked against it.
 */
#ifndef GAZ_PUCT_H
#define GAZ_PUCT_H
#include <stdint.h>
#include "gaz_det.h"
#include "gaz_games.h"

#ifdef __cplusplus
extern "C" {
#endif

/* evaluator = the reference's duck-typed session.run (MCTS.py:224-235): int8 state
 * [H][W][C] in (the reference casts to float32; values are small integers), policy
 * f32[A] and value f32 out.  depth = the Cache_Wrapper depth argument (unused by oracle). */
typedef void (*gaz_eval_fn)(void* ctx, const int8_t* state, int n_state, float* policy, float* value);

#define GAZ_NOT_TERMINAL (-9) /* Python None in Node.is_terminal */

typedef struct gaz_node {
    struct gaz_node* parent;
    int child_id;
    int8_t* board;            /* NULL once deleted / never kept */
    int* history;             /* action_history (full, from move 0) */
    int n_history;
    int current_player;
    struct gaz_node** children; int n_children;   /* expanded children, in prior order */
    int* legal_actions;       /* child actions sorted by descending prior; [n_children] is next to pop */
    int n_actions;            /* len(child_visits) */
    uint32_t* child_visits;
    float* child_values;
    float* child_prob_priors;
    int is_terminal;          /* GAZ_NOT_TERMINAL, or winner (-1/1) / 0 draw */
    int shares_stats;         /* stats arrays owned elsewhere (not used: we re-root in place) */
} gaz_node;

typedef struct {
    gaz_game_desc g;
    /* live game the tree is attached to (MCTS holds a reference to the game object) */
    const int8_t* game_board; const int* game_history; const int* game_n_history; const int* game_next_player;
    gaz_eval_fn eval; void* eval_ctx;
    double c_puct_init, c_puct_base;
    int use_dirichlet; double dirichlet_alpha, dirichlet_epsilon;
    double tau;
    int fast_find_win;
    gaz_node* root; uint64_t root_visits;
    gaz_event ev;             /* RNG stream of this tree; ev.event advances per random call */
    uint64_t n_evals;         /* evaluator calls made (for the evals/move measurement) */
    uint64_t n_nodes;
} gaz_puct;

/* one row of MCTS.run's move_probs (MCTS.py:591-600) */
typedef struct {
    int action; double prob; double winrate; float value; uint32_t visits; float prior;
    uint64_t root_visits; int is_terminal;
} gaz_move_row;

gaz_puct* gaz_puct_create(int game_id, const int8_t* game_board, const int* game_history, const int* game_n_history,
                          const int* game_next_player, gaz_eval_fn eval, void* ctx,
                          double c_puct_init, double c_puct_base, int use_dirichlet, double alpha, double eps,
                          double tau, uint64_t seed, uint32_t slot, uint32_t game_seq, uint32_t tree);
void gaz_puct_destroy(gaz_puct* t);
void gaz_puct_set_tau(gaz_puct* t, double tau);
/* MCTS.run (MCTS.py:528-618): returns chosen action; rows (unsorted, child order) into out_rows, count in *n_rows */
int gaz_puct_run(gaz_puct* t, int iteration_limit, gaz_move_row* out_rows, int* n_rows);
/* MCTS.prune_tree (MCTS.py:657-671) */
void gaz_puct_prune(gaz_puct* t, int action, int create_new_root);

/* K1 exposed for micro-fixtures: MCTS._get_best_PUCT_score_index (MCTS.py:172-191) */
int gaz_puct_best_index(const float* priors, const float* values, const uint32_t* visits, int n,
                        uint64_t parent_visits, double c_init, double c_base, int use_libm);

#ifdef __cplusplus
}
#endif
#endif
