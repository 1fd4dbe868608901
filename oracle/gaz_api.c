/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gaz_det.h header).
 * gaz_api.c — flat entry points for ctypes (tests, fixture generator): the injected
 * noise stream and the numeric helpers, so Python draws exactly what the C oracle draws.
 */
#include "gaz_det.h"
#include "gaz_games.h"

static gaz_event mk(uint64_t seed, uint32_t slot, uint32_t seq, uint32_t tree, uint32_t event, uint32_t purpose) {
    gaz_event e; e.key[0] = (uint32_t)seed; e.key[1] = (uint32_t)(seed >> 32);
    e.slot = slot; e.game_seq = seq; e.event = event; e.tree = tree; e.purpose = purpose; return e;
}
void gaz_api_dirichlet(uint64_t seed, uint32_t slot, uint32_t seq, uint32_t tree, uint32_t event, double alpha, int n, double* out) {
    gaz_event e = mk(seed, slot, seq, tree, event, GAZ_P_DIRICHLET); gaz_dirichlet(&e, alpha, n, out);
}
uint32_t gaz_api_pick(uint64_t seed, uint32_t slot, uint32_t seq, uint32_t tree, uint32_t event, uint32_t n) {
    gaz_event e = mk(seed, slot, seq, tree, event, GAZ_P_TERMINAL_PICK); return gaz_pick(&e, n);
}
double gaz_api_uniform(uint64_t seed, uint32_t slot, uint32_t seq, uint32_t tree, uint32_t event, uint32_t purpose) {
    gaz_event e = mk(seed, slot, seq, tree, event, purpose); return gaz_uniform(&e);
}
void gaz_api_gumbel(uint64_t seed, uint32_t slot, uint32_t seq, uint32_t tree, uint32_t event, int n, double* out) {
    gaz_event e = mk(seed, slot, seq, tree, event, GAZ_P_GUMBEL);
    for (int i = 0; i < n; ++i) out[i] = gaz_gumbel(&e, (uint32_t)i);
}
void gaz_api_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { gaz_philox(ctr, key, out); }
double gaz_api_log(double x) { return gaz_log(x); }
double gaz_api_exp(double x) { return gaz_exp(x); }
float gaz_api_np_sum_f32(const float* a, int n) { return gaz_np_sum_f32(a, n); }
double gaz_api_np_sum_f64(const double* a, int n) { return gaz_np_sum_f64(a, n); }

/* game rule entry points for the Game_Tester-style fixtures */
int gaz_api_legal_actions(int game, const int8_t* board, int* out) { gaz_game_desc g = gaz_game(game); return gaz_legal_actions(&g, board, out); }
void gaz_api_do_action(int game, int8_t* board, int action, int player) { gaz_game_desc g = gaz_game(game); gaz_do_action(&g, board, action, player); }
int gaz_api_check_win(int game, const int8_t* board, int player, int last_action) { gaz_game_desc g = gaz_game(game); return gaz_check_win(&g, board, player, last_action); }
void gaz_api_input_state(int game, const int8_t* board, int current_player, const int* hist, int n_hist, int8_t* out) {
    gaz_game_desc g = gaz_game(game); gaz_input_state(&g, board, current_player, hist, n_hist, out);
}
