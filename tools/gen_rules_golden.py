"""Generate tests/golden/rules_<game>.npz from the REFERENCE's own Game classes (Connect4/Connect4.py, Gomoku/Gomoku.py,
TicTacToe/Tictactoe.py), imported under tools/ref_shim.py in the build container only — the Game_Tester-style rule fixtures
SURVEY 8c asks for (Game_Tester.py:77-125,297-405).  Playouts are driven here (own seeded generator picks the moves); everything
recorded is an output of the reference: per position (= every prefix of every playout, the finished one included)

    hist / n_hist   the action indices played from the empty board (Connect4: column; others: y * W + x)
    board           game.board
    legal           get_legal_actions() as a mask over the policy index
    winner          check_win() (== check_win_MCTS(board, -next_player, history), asserted here as Game_Tester does)
    input           get_input_state()
    policy_in       a random positive float32 policy vector (input, drawn here)
    legal_policy    get_legal_actions_policy_MCTS(board, -next_player, history, policy_in, normalize=True), scattered by action
    terminal        MCTS.get_terminal_actions_fn(do_action_MCTS, check_win_MCTS, ...) for the side to move: -1 / 0 draws / 1 wins
                    per action (all -1 once the game is over: the search never expands a finished position)

Data only; no reference source.      python tools/gen_rules_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tools import ref_shim  # noqa: E402
from tools.gen_golden import GAME_CLASS, action_to_index  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
PLAYOUTS = {"TicTacToe": (60, 0.3), "Connect4": (48, 0.35), "Gomoku": (10, 0.6)}     # playouts, P(take a winning move when there is one)
DRAW_SEEKING = {"Connect4": 2, "TicTacToe": 0, "Gomoku": 0}   # extra playouts steered away from wins until the board is full (a draw)
POLICY_STRIDE = {"TicTacToe": 1, "Connect4": 1, "Gomoku": 4}  # policy vectors are recorded for every n-th position (fixture size)
MAX_PLIES = {"TicTacToe": 9, "Connect4": 42, "Gomoku": 225}


def future_histories(game, legal):
    ah = np.array(game.action_history)
    if game.action_history:
        rep = np.zeros((len(legal), *ah.shape), dtype=ah.dtype); rep[:] = ah
        return np.concatenate((rep, np.expand_dims(legal, 1)), axis=1)          # MCTS.py:301-307
    return np.expand_dims(legal, 1)


def snapshot(ref, name, game, rng, A):
    cls = type(game)
    hist = [action_to_index(name, a) for a in game.action_history]
    legal = game.get_legal_actions()
    mask = np.zeros(A, np.uint8)
    for a in legal:
        mask[action_to_index(name, a)] = 1
    w = int(game.check_win()) if game.action_history else -2
    if game.action_history:
        assert w == int(cls.check_win_MCTS(game.board, -game.next_player, np.array(game.action_history)))      # Game_Tester.py:351-360
    pol_in = (rng.random(A, dtype=np.float32) + np.float32(1e-3)).astype(np.float32)
    lp = np.zeros(A, np.float32)
    if len(legal):
        la, lpv = cls.get_legal_actions_policy_MCTS(game.board, -game.next_player, np.array(game.action_history), pol_in.copy(), normalize=True, shuffle=False)
        assert len(la) == len(legal)
        for a, p in zip(la, lpv):
            lp[action_to_index(name, a)] = p
    term = np.full(A, -1, np.int32)
    wins = []
    if w == -2 and len(legal):
        ta, tm = ref["MCTS"].MCTS.get_terminal_actions_fn(cls.do_action_MCTS, cls.check_win_MCTS, future_histories(game, legal), game.board,
                                                         game.next_player, False)
        for a, m in zip(ta, tm):
            term[action_to_index(name, a)] = int(m)
            if m == 1.0:
                wins.append(a)
    return dict(hist=hist, board=game.board.copy(), legal=mask, winner=w, input=np.array(game.get_input_state()).copy(), policy_in=pol_in,
                legal_policy=lp, terminal=term), legal, wins


def main():
    ref = ref_shim.load_reference()
    for name, (n_playouts, p_win) in PLAYOUTS.items():
        rng = np.random.default_rng(20260 + len(name))
        cls = getattr(ref[GAME_CLASS[name][0]], GAME_CLASS[name][1])
        rows = []
        ends = {-1: 0, 0: 0, 1: 0, "cut": 0}
        def safe_moves(g, legal):
            """moves that neither win nor leave the opponent a winning reply (1-ply look-ahead with the reference's own functions)"""
            out = []
            for a in legal:
                b2 = cls.do_action_MCTS(g.board.copy(), a, g.next_player)
                h2 = np.array(list(g.action_history) + [a])
                if cls.check_win_MCTS(b2, g.next_player, h2) != -2:
                    continue
                opp_legal = cls.get_legal_actions_MCTS(b2, g.next_player, h2)
                if any(cls.check_win_MCTS(cls.do_action_MCTS(b2.copy(), o, -g.next_player), -g.next_player, np.array(list(h2) + [o])) == -g.next_player
                       for o in opp_legal):
                    continue
                out.append(a)
            return out

        draws_wanted, attempts = DRAW_SEEKING[name], 0
        k = 0
        while k < n_playouts + draws_wanted:
            seeking = k >= n_playouts
            g = cls(); A = g.policy_shape[0]
            game_rows = []
            while True:
                snap, legal, wins = snapshot(ref, name, g, rng, A)
                game_rows.append(snap)
                if snap["winner"] != -2:
                    break
                if len(legal) == 0 or len(g.action_history) >= MAX_PLIES[name]:
                    break
                if seeking:
                    nonwin = [a for a in legal if not any(np.array_equal(a, w_) for w_ in wins)]
                    cand = safe_moves(g, legal) or nonwin or list(legal)
                    a = cand[int(rng.integers(len(cand)))]
                elif wins and rng.random() < p_win:
                    a = wins[int(rng.integers(len(wins)))]
                elif name == "Gomoku" and g.action_history and rng.random() < 0.85:
                    # stay near the stones already played so that lines (and five-in-a-rows) actually form
                    last = np.array(g.action_history[int(rng.integers(max(0, len(g.action_history) - 6), len(g.action_history)))], np.int64)
                    near = [a for a in legal if max(abs(int(a[0]) - last[0]), abs(int(a[1]) - last[1])) <= 1]
                    a = near[int(rng.integers(len(near)))] if near else legal[int(rng.integers(len(legal)))]
                else:
                    a = legal[int(rng.integers(len(legal)))]
                g.do_action(a)
            last = game_rows[-1]
            if seeking and last["winner"] != 0:
                attempts += 1
                assert attempts < 2000, "no drawn game found"
                continue                                                # not a draw: try again
            rows += game_rows
            ends[last["winner"] if last["winner"] != -2 else "cut"] += 1
            k += 1
        P = len(rows); maxT = max(len(r["hist"]) for r in rows)
        hist = np.full((P, max(maxT, 1)), -1, np.int32)
        for i, r in enumerate(rows):
            hist[i, :len(r["hist"])] = r["hist"]
        out = dict(game=name, hist=hist, n_hist=np.array([len(r["hist"]) for r in rows], np.int32))
        for key in ("board", "legal", "winner", "input", "terminal"):
            out[key] = np.array([r[key] for r in rows])
        pr = np.arange(0, P, POLICY_STRIDE[name], dtype=np.int32)
        out["policy_rows"] = pr
        out["policy_in"] = np.array([rows[i]["policy_in"] for i in pr]); out["legal_policy"] = np.array([rows[i]["legal_policy"] for i in pr])
        np.savez_compressed(os.path.join(GOLD, f"rules_{name.lower()}.npz"), **out)
        n_term = int((out["terminal"] >= 0).any(1).sum())
        print(name, "positions", P, "ends", ends, "positions with terminal moves", n_term, "dtype board", out["board"].dtype, "input", out["input"].shape, flush=True)


if __name__ == "__main__":
    main()
