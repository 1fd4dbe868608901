/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gaz_det.h header).
 *
 * gaz_games.h — CPU restatement of the three complete Game plugins' static
 * *_MCTS functions (the ones the search calls):
 *   Connect4  /root/reference/Connect4/Connect4.py:269-411
 *   Gomoku    /root/reference/Gomoku/Gomoku.py:112-255
 *   TicTacToe /root/reference/TicTacToe/Tictactoe.py:184-300
 *
 * Boards are int8 row-major [H][W] exactly like the reference's numpy boards.
 * Actions are carried as one int: Connect4 = column x (the reference's int8
 * scalar); Gomoku / TicTacToe = y*W + x for the reference's [x, y] pair, which
 * is also the policy index (board.reshape(-1) order, Gomoku.py:129-138,
 * Tictactoe.py:198-199).
 */
#ifndef GAZ_GAMES_H
#define GAZ_GAMES_H
#include <stdint.h>
#include <string.h>

enum { GAZ_TTT = 0, GAZ_C4 = 1, GAZ_GMK = 2 };
#define GAZ_RUNNING (-2)

typedef struct {
    int id, H, W, C;  /* C = channels of the NN input state */
    int A;            /* policy_shape[0] */
} gaz_game_desc;

static inline gaz_game_desc gaz_game(int id) {
    gaz_game_desc g;
    g.id = id;
    if (id == GAZ_TTT) { g.H = 3; g.W = 3; g.C = 2; g.A = 9; }
    else if (id == GAZ_C4) { g.H = 6; g.W = 7; g.C = 4; g.A = 7; }
    else { g.H = 15; g.W = 15; g.C = 2; g.A = 225; }
    return g;
}

/* get_legal_actions_MCTS: Connect4.py:271-276 (columns whose |sum| < 6, ascending x),
 * Gomoku.py:114-119 / Tictactoe.py:186-187 (np.argwhere(board == 0), row-major). */
static inline int gaz_legal_actions(const gaz_game_desc* g, const int8_t* board, int* out) {
    int n = 0;
    if (g->id == GAZ_C4) {
        for (int x = 0; x < 7; ++x) {
            int s = 0;
            for (int y = 0; y < 6; ++y) s += board[y * 7 + x] < 0 ? -board[y * 7 + x] : board[y * 7 + x];
            if (s < 6) out[n++] = x;
        }
    } else {
        for (int i = 0; i < g->H * g->W; ++i) if (board[i] == 0) out[n++] = i;
    }
    return n;
}

/* do_action_MCTS: Connect4.py:311-316 (drop into row 5 - sum|col|), Gomoku.py:164-167,
 * Tictactoe.py:221-224 (board[y][x] = player). */
static inline void gaz_do_action(const gaz_game_desc* g, int8_t* board, int action, int player) {
    if (g->id == GAZ_C4) {
        int s = 0;
        for (int y = 0; y < 6; ++y) s += board[y * 7 + action] != 0;
        board[(5 - s) * 7 + action] = (int8_t)player;
    } else {
        board[action] = (int8_t)player;
    }
}

/* check_win_MCTS(board, current_player, action_history): returns current_player,
 * 0 (draw) or -2 (running).  Only action_history[-1] is read by the reference. */
static inline int gaz_check_win(const gaz_game_desc* g, const int8_t* board, int current_player, int last_action) {
    if (g->id == GAZ_C4) { /* Connect4.py:353-411 */
        int x = last_action, y = -1;
        for (int r = 0; r < 6; ++r) if (board[r * 7 + x] == current_player) { y = r; break; } /* min(where(col==player)) */
        int start_x = x - 3 > 0 ? x - 3 : 0, end_x = x + 3 < 6 ? x + 3 : 6;
        int start_y = y + 3 < 5 ? y + 3 : 5, end_y = y - 3 > 0 ? y - 3 : 0;
        int count = 0;
        for (int i = start_x; i <= end_x; ++i) {
            if (board[y * 7 + i] == current_player) { if (++count == 4) return current_player; } else count = 0;
        }
        count = 0;
        for (int i = start_y; i >= end_y; --i) {
            if (board[i * 7 + x] == current_player) { if (++count == 4) return current_player; } else count = 0;
        }
        int lo, hi;
        count = 0; /* bottom-left -> top-right: cells (y - i, x + i) */
        lo = -((x - start_x) < (start_y - y) ? (x - start_x) : (start_y - y));
        hi = (end_x - x) < (y - end_y) ? (end_x - x) : (y - end_y);
        for (int i = lo; i <= hi; ++i) {
            if (board[(y - i) * 7 + x + i] == current_player) { if (++count == 4) return current_player; } else count = 0;
        }
        count = 0; /* top-left -> bottom-right: cells (y + i, x + i) */
        lo = -((x - start_x) < (y - end_y) ? (x - start_x) : (y - end_y));
        hi = (end_x - x) < (start_y - y) ? (end_x - x) : (start_y - y);
        for (int i = lo; i <= hi; ++i) {
            if (board[(y + i) * 7 + x + i] == current_player) { if (++count == 4) return current_player; } else count = 0;
        }
        for (int i = 0; i < 42; ++i) if (board[i] == 0) return GAZ_RUNNING;
        return 0;
    } else if (g->id == GAZ_GMK) { /* Gomoku.py:194-255; never returns a draw */
        int cx = last_action % 15, cy = last_action / 15;
        static const int dxs[4] = {1, 0, 1, -1}, dys[4] = {0, 1, 1, 1};
        for (int d = 0; d < 4; ++d) {
            int fives = 0;
            for (int i = -4; i < 5; ++i) {
                int nx = cx + dxs[d] * i, ny = cy + dys[d] * i;
                if (nx >= 0 && nx <= 14 && ny >= 0 && ny <= 14) {
                    if (board[ny * 15 + nx] == current_player) { if (++fives == 5) return current_player; } else fives = 0;
                }
            }
        }
        return GAZ_RUNNING;
    } else { /* Tictactoe.py:275-300: any full line of equal non-zero marks */
        for (int r = 0; r < 3; ++r)
            if (board[r * 3] != 0 && board[r * 3] == board[r * 3 + 1] && board[r * 3 + 1] == board[r * 3 + 2]) return current_player;
        for (int c = 0; c < 3; ++c)
            if (board[c] != 0 && board[c] == board[3 + c] && board[3 + c] == board[6 + c]) return current_player;
        if (board[0] != 0 && board[0] == board[4] && board[4] == board[8]) return current_player;
        if (board[2] != 0 && board[2] == board[4] && board[4] == board[6]) return current_player;
        for (int i = 0; i < 9; ++i) if (board[i] == 0) return GAZ_RUNNING;
        return 0;
    }
}

/* get_input_state_MCTS(board, current_player, action_history) -> int8 [H][W][C].
 * hist = full action history of the position, n_hist its length.
 * Connect4.py:329-346: planes (before the final transpose) are
 *   [0] = current_player, [3] = board, [2] = board minus last move, [1] = minus last two,
 *   and with >= 4 moves played plane [0] is OVERWRITTEN by the board minus the last three
 *   (board_state[i - 1] with i = -3 is index 0) — reproduced as is.
 * Gomoku.py:175-177 / Tictactoe.py:231-235: planes (-current_player, board). */
static inline void gaz_input_state(const gaz_game_desc* g, const int8_t* board, int current_player,
                                   const int* hist, int n_hist, int8_t* out) {
    int HW = g->H * g->W;
    if (g->id == GAZ_C4) {
        int8_t planes[4][42];
        memset(planes, 0, sizeof(planes));
        for (int i = 0; i < 42; ++i) { planes[0][i] = (int8_t)current_player; planes[3][i] = board[i]; }
        int max_length = n_hist - 1; if (max_length > 3) max_length = 3;
        int8_t prev[42]; memcpy(prev, board, 42);
        for (int i = 1; i <= max_length; ++i) { /* python i = -1 .. -max_length */
            int x = hist[n_hist - i];
            int y = 0; while (prev[y * 7 + x] == 0) ++y; /* min(where(prev[:, x] != 0)) */
            prev[y * 7 + x] = 0;
            memcpy(planes[3 - i], prev, 42); /* board_state[-i - 1] */
        }
        for (int i = 0; i < 42; ++i) for (int c = 0; c < 4; ++c) out[i * 4 + c] = planes[c][i];
    } else {
        for (int i = 0; i < HW; ++i) { out[i * 2] = (int8_t)(-current_player); out[i * 2 + 1] = board[i]; }
    }
}

/* policy index of an action (get_legal_actions_policy_MCTS gather). */
static inline int gaz_policy_index(const gaz_game_desc* g, int action) { (void)g; return action; }

#endif
