/* ORACLE — TEST INFRASTRUCTURE ONLY (see gaz_det.h header). */
#ifndef GAZ_SELFPLAY_H
#define GAZ_SELFPLAY_H
#include <stdint.h>
#include "gaz_puct.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint32_t salt; int A; } gaz_hash_eval_ctx;
void gaz_hash_eval(void* ctx, const int8_t* state, int n_state, float* policy, float* value);

/* the train_config keys Self_Play reads on the PUCT path (Self_Play.py:34-57,79-157) */
typedef struct {
    int game_id;
    int run_iterations;           /* int(MCTS_iteration_limit * 1.5), Self_Play.py:99 */
    int max_actions;
    int num_explore_actions_first, num_explore_actions_second;
    double c_puct_init, c_puct_base, dirichlet_alpha;
    int create_new_root;
    int n_opening;                /* train_config["opening_actions"] (Self_Play.py:130-140): [action, weight] pairs for move 0 */
    int opening_actions[8]; double opening_weights[8];
} gaz_sp_config;

/* the override of move 0: returns the action to play given the search's own choice */
int gaz_opening_override(const gaz_sp_config* cfg, int mcts_action, uint64_t seed, uint32_t slot, uint32_t game_seq);

/* caller-allocated record of one game, cap_T plies */
typedef struct {
    int cap_T, T, winner;
    int8_t* states;      /* [T][H*W*C]   Self_Play.py:80 */
    float* policies;     /* [T][A]       Self_Play.py:114-115 */
    float* q; float* z; float* values;   /* [T]  Self_Play.py:117-127,165-172 */
    int* actions;        /* [T] */
    uint32_t* root_N; float* root_W; float* root_P;   /* [T][A], indexed by policy index of the root child's action */
    uint64_t* root_visits;               /* [T] */
    uint32_t* evals;                     /* [T] evaluator calls made by the running tree's run() */
    uint64_t total_evals;
} gaz_sp_record;

int gaz_selfplay_game(const gaz_sp_config* cfg, gaz_eval_fn eval, void* ctx, uint64_t seed, uint32_t slot,
                      uint32_t game_seq, gaz_sp_record* rec);
void gaz_oracle_set_libm(int on);
/* Self_Play.play with use_gumbel = True: MCTS_Gumbel.run(iteration_limit) per move, fresh tree every move */
int gaz_selfplay_game_gumbel(const gaz_sp_config* cfg, int m, double c_visit, double c_scale, int iteration_limit,
                             gaz_eval_fn eval, void* ctx, uint64_t seed, uint32_t slot, uint32_t game_seq, int use_libm,
                             gaz_sp_record* rec);
#ifdef __cplusplus
}
#endif
#endif
