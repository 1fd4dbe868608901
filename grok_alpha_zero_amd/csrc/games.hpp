// games.hpp — device twins of the reference's Game plugins (the static *_MCTS functions the search calls).
//
//   Connect4  /root/reference/Connect4/Connect4.py:269-411
//   Gomoku    /root/reference/Gomoku/Gomoku.py:112-255
//   TicTacToe /root/reference/TicTacToe/Tictactoe.py:184-300
//
// Boards are int8 row-major [H][W] (player -1 / +1, 0 empty), as in the reference.  An action is one
// byte: Connect4 the column x; Gomoku / TicTacToe y*W + x (= the policy index, board.reshape(-1) order).
// Functions are per-lane: the caller fans lanes out over candidate actions / cells.
#pragma once
#include "wave.hpp"

namespace gaz {

enum : int { GAME_TTT = 0, GAME_C4 = 1, GAME_GMK = 2 };
enum : int { RUNNING = -2 };

template <int ID> struct Game;

template <> struct Game<GAME_TTT> {
    static constexpr int ID = GAME_TTT, H = 3, W = 3, HW = 9, C = 2, A = 9, K = 3;
    static constexpr int APAD = 16, BPAD = 16, MAXT = 9, TPAD = 16;
    static constexpr bool DRAWS = true;
    static constexpr int TEAM = GAZ_TEAM(64);       // lanes that own one game (wave.hpp "Teams")
};
template <> struct Game<GAME_C4> {
    static constexpr int ID = GAME_C4, H = 6, W = 7, HW = 42, C = 4, A = 7, K = 4;
    static constexpr int APAD = 8, BPAD = 48, MAXT = 42, TPAD = 48;
    static constexpr bool DRAWS = true;
    static constexpr int TEAM = GAZ_TEAM(64);
};
template <> struct Game<GAME_GMK> {
    static constexpr int ID = GAME_GMK, H = 15, W = 15, HW = 225, C = 2, A = 225, K = 5;
    static constexpr int APAD = 232, BPAD = 240, MAXT = 225, TPAD = 232;
    static constexpr bool DRAWS = false;   // check_win_MCTS never reports a draw (Gomoku.py:249-255)
    static constexpr int TEAM = GAZ_TEAM(64);
};

// The same game stepped by a 16-lane team: FOUR games per wavefront (PUCT tree kernel of the small boards).  Identical rules and
// record layouts — only the lane mapping differs, so HBM state written by one variant is read by the other.
#ifndef GAZ_TEAM_LANES
#define GAZ_TEAM_LANES 16
#endif
template <int ID> struct TeamGame : Game<ID> { static constexpr int TEAM = GAZ_TEAM(GAZ_TEAM_LANES); };
template <class G> struct PuctVariant { typedef G type; };
template <> struct PuctVariant<Game<GAME_TTT>> { typedef TeamGame<GAME_TTT> type; };
template <> struct PuctVariant<Game<GAME_C4>> { typedef TeamGame<GAME_C4> type; };

// is action a legal on this board?  (get_legal_actions_MCTS: Connect4.py:271-276 column not full —
// pieces stack from row 5 upward so "sum |col| < 6" == top cell empty; Gomoku.py:114-119 /
// Tictactoe.py:186-187 empty cell)
template <class G> GAZ_DEV bool action_legal(const int8_t* board, int a) {
    if (G::ID == GAME_C4) return board[a] == 0;
    return board[a] == 0;
}

// cell a stone lands on (do_action_MCTS: Connect4.py:311-316 row 5 - #pieces in the column)
template <class G> GAZ_DEV int landing_cell(const int8_t* board, int a) {
    if (G::ID == GAME_C4) {
        int cnt = 0;
#pragma unroll
        for (int y = 0; y < 6; ++y) cnt += board[y * 7 + a] != 0;
        return (5 - cnt) * 7 + a;
    }
    return a;
}

// would `player` placing a stone on empty `cell` complete a line of K?  Equivalent to check_win_MCTS on
// the board after the move (Connect4.py:353-406 4-in-a-row windows through the last stone; Gomoku.py:199-248
// +-4 windows, overlines count; Tictactoe.py:276-291 any full line — earlier positions have none).
template <class G> GAZ_DEV bool wins_after(const int8_t* board, int cell, int player) {
    const int y = cell / G::W, x = cell % G::W;
    const int dxs[4] = {1, 0, 1, 1}, dys[4] = {0, 1, 1, -1};
    bool win = false;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int run = 1;
#pragma unroll
        for (int s = 1; s < G::K; ++s) {
            int nx = x + dxs[d] * s, ny = y + dys[d] * s;
            if (nx < 0 || nx >= G::W || ny < 0 || ny >= G::H || board[ny * G::W + nx] != player) break;
            ++run;
        }
#pragma unroll
        for (int s = 1; s < G::K; ++s) {
            int nx = x - dxs[d] * s, ny = y - dys[d] * s;
            if (nx < 0 || nx >= G::W || ny < 0 || ny >= G::H || board[ny * G::W + nx] != player) break;
            ++run;
        }
        win = win || (run >= G::K);
    }
    return win;
}

}  // namespace gaz
