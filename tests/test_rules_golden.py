"""Game rules against fixtures recorded from the REFERENCE's own Game classes (tools/gen_rules_golden.py: random playouts of
Connect4/Connect4.py, Gomoku/Gomoku.py, TicTacToe/Tictactoe.py that run to wins, draws and full boards; Game_Tester.py:77-125,
297-405).  Three rule implementations are held to the same data:
  (i)   the oracle's C rules (oracle/gaz_games.h) — CPU suite
  (ii)  the host plugins (grok_alpha_zero_amd/games.py) — CPU suite
  (iii) the DEVICE rule code the search kernels call (csrc/games.hpp + puct_core.hpp), through gaz_engine_probe_rules: on the
        one-lane emulation build in the CPU suite and on the MI355X under -m gpu.
Bit-exact: boards, legal sets, winners, input planes, terminal-move classification; normalised legal policies are float32 and
compared exactly too (numpy's pairwise float32 sum order is reproduced)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from grok_alpha_zero_amd.games import GAMES

GID = {"TicTacToe": 0, "Connect4": 1, "Gomoku": 2}
NAMES = ["TicTacToe", "Connect4", "Gomoku"]


def _fx(name):
    return np.load(os.path.join(GOLDEN, f"rules_{name.lower()}.npz"))


def _hist(fx, i):
    return fx["hist"][i, :fx["n_hist"][i]].tolist()


def _i8(a):
    return a.ctypes.data_as(C.POINTER(C.c_int8))


@pytest.mark.parametrize("name", NAMES)
def test_fixture_covers_natural_game_ends(name):
    fx = _fx(name)
    w = fx["winner"]
    assert (w == -1).any() and (w == 1).any() and (w == -2).sum() > 100
    if name != "Gomoku":                                   # Gomoku's check_win never reports a draw (Gomoku.py:249-255)
        assert (w == 0).any()
        full = [i for i in range(len(w)) if (fx["board"][i] != 0).all()]
        assert full and all(w[i] != -2 for i in full)      # a full board is never "still running"
    assert ((fx["terminal"] == 1).any(1)).sum() >= 10      # positions with a winning move for the side to move


@pytest.mark.parametrize("name", NAMES)
def test_oracle_c_rules_match_reference_game_classes(oracle, name):
    L = oracle.lib()
    fx = _fx(name)
    gid = GID[name]
    for i in range(len(fx["winner"])):
        h = _hist(fx, i)
        board = np.zeros(fx["board"][i].size, np.int8)
        player, winner = -1, -2
        for a in h:
            L.gaz_api_do_action(gid, _i8(board), a, player)
            winner = L.gaz_api_check_win(gid, _i8(board), player, a)
            player = -player
        assert np.array_equal(board.reshape(fx["board"][i].shape), fx["board"][i])
        assert winner == fx["winner"][i], (i, h)
        buf = (C.c_int * 256)()
        k = L.gaz_api_legal_actions(gid, _i8(board), buf)
        mask = np.zeros(fx["legal"].shape[1], np.uint8); mask[list(buf[:k])] = 1
        assert np.array_equal(mask, fx["legal"][i])
        out = np.zeros(fx["input"][i].size, np.int8); hh = np.array(h, np.int32)
        L.gaz_api_input_state(gid, _i8(board), -player, hh.ctypes.data_as(C.POINTER(C.c_int)), len(h), _i8(out))
        assert np.array_equal(out.reshape(fx["input"][i].shape), fx["input"][i]), (i, h)


@pytest.mark.parametrize("name", NAMES)
def test_host_plugins_match_reference_game_classes(name):
    fx = _fx(name)
    cls = GAMES[name]
    pol_of = {int(r): k for k, r in enumerate(fx["policy_rows"])}
    for i in range(len(fx["winner"])):
        g = cls()
        for a in _hist(fx, i):
            g.do_action(cls.index_to_action(a))
        assert np.array_equal(g.board, fx["board"][i]) and g.board.dtype == fx["board"].dtype
        legal = g.get_legal_actions()
        mask = np.zeros(fx["legal"].shape[1], np.uint8)
        for a in legal:
            mask[cls.action_to_index(a)] = 1
        assert np.array_equal(mask, fx["legal"][i])
        assert (g.check_win() if g.action_history else -2) == fx["winner"][i]
        assert np.array_equal(np.asarray(g.get_input_state()), fx["input"][i])
        if i in pol_of and len(legal):
            la, lp = cls.get_legal_actions_policy_MCTS(g.board, -g.next_player, np.array(g.action_history), fx["policy_in"][pol_of[i]].copy())
            got = np.zeros(mask.size, np.float32)
            for a, p in zip(la, lp):
                got[cls.action_to_index(a)] = p
            np.testing.assert_array_equal(got, fx["legal_policy"][pol_of[i]])


def _check_device(name, lib_path):
    from grok_alpha_zero_amd.engine import SelfPlayEngine
    fx = _fx(name)
    eng = SelfPlayEngine(name, 2, 8, fx["hist"].shape[1] if name != "Gomoku" else 225, 0, 0, 2.5, 0.5, seed=1, lib_path=lib_path)
    P = len(fx["winner"])
    hists = [_hist(fx, i) for i in range(P)]
    r = eng.probe_rules(hists)
    np.testing.assert_array_equal(r["board"], fx["board"])
    np.testing.assert_array_equal(r["winner"], fx["winner"])
    np.testing.assert_array_equal(r["legal"].astype(np.uint8), fx["legal"])
    np.testing.assert_array_equal(r["input"], fx["input"])
    np.testing.assert_array_equal(r["terminal"], fx["terminal"])
    rows = fx["policy_rows"]
    rp = eng.probe_rules([hists[i] for i in rows], policy=fx["policy_in"])
    np.testing.assert_array_equal(rp["legal_policy"], fx["legal_policy"])
    eng.close()


@pytest.mark.parametrize("name", NAMES)
def test_device_rules_match_reference_game_classes_emu(name):
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu_dir])
    _check_device(name, os.path.join(emu_dir, "libgaz_emu.so"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_rules_match_reference_game_classes_hip(name):
    """The same comparison through libgaz_engine.so on the MI355X: the wave-parallel rule code (ballots, one lane per candidate)."""
    import torch
    assert torch.cuda.is_available()
    _check_device(name, None)
