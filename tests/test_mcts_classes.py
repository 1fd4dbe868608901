"""CPU suite: the drop-in search classes (grok_alpha_zero_amd/mcts.py: MCTS, MCTS_Gumbel with the reference's constructor /
run / prune_tree surface) on the emulation build, against fixtures recorded by driving the REFERENCE's classes the same way
(single tree playing both sides, tools/gen_golden.py: ref_single_tree_puct) and against the Self_Play Gumbel fixtures."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from grok_alpha_zero_amd.games import GAMES

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU = os.path.join(EMU_DIR, "libgaz_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return EMU


class _HashSession:
    def __init__(self, oracle, A, salt):
        self.o, self.A, self.salt, self.calls = oracle, A, salt, 0

    def run(self, output_names, input_feed, **kw):
        self.calls += 1
        pol, val = self.o.hash_eval(input_feed["inputs"][0].astype(np.int8), self.A, self.salt)
        return [pol.reshape(1, -1), np.array([[val]], np.float32)]


@pytest.mark.parametrize("name", ["c4_mcts_single", "ttt_mcts_single", "c4_mcts_single_ffw", "ttt_mcts_single_ffw"])
@pytest.mark.parametrize("with_session", [True, False], ids=["session", "builtin"])
def test_mcts_class_matches_reference_class(emu_lib, oracle, name, with_session):
    from grok_alpha_zero_amd.mcts import MCTS
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    game = GAMES[str(fx["game"])]()
    A = game.policy_shape[0]
    sess = _HashSession(oracle, A, int(fx["salt"])) if with_session else None
    mcts = MCTS(game, sess, c_puct_init=float(fx["c_puct_init"]), use_dirichlet=True, dirichlet_alpha=float(fx["dirichlet_alpha"]),
                dirichlet_epsilon=0.25, tau=1.0, fast_find_win=bool(int(fx["fast_find_win"])) if "fast_find_win" in fx else False,
                seed=int(fx["seed"]), hash_salt=int(fx["salt"]), lib_path=emu_lib)
    for ply in range(len(fx["actions"])):
        mcts.update_hyperparams(tau=1.0 if ply < 4 else 0)
        move, rows = mcts.run(iteration_limit=int(fx["iteration_limit"]), use_bar=False)
        N = np.zeros(A, np.uint32); Wv = np.zeros(A, np.float32)
        for r in rows:
            a = type(game).action_to_index(r[0]); N[a] = r[4]; Wv[a] = r[3]
        assert type(game).action_to_index(move) == fx["actions"][ply]
        np.testing.assert_array_equal(N, fx["root_N"][ply])
        np.testing.assert_array_equal(Wv, fx["root_W"][ply])
        assert rows[0][6] == fx["root_visits"][ply] and rows == sorted(rows, key=lambda r: r[4], reverse=True)
        game.do_action(move)
        if game.check_win() != -2:
            break
        mcts.prune_tree(move)
    assert ply == len(fx["actions"]) - 1
    if with_session:
        assert sess.calls == int(fx["evaluator_calls"])
    mcts.close()


def test_mcts_gumbel_class_plays_the_self_play_fixture(emu_lib, oracle):
    from grok_alpha_zero_amd.mcts import MCTS_Gumbel
    fx = np.load(os.path.join(GOLDEN, "c4_gumbel_a.npz"))
    game = GAMES["Connect4"]()
    sess = _HashSession(oracle, 7, int(fx["salt"]))
    mcts = MCTS_Gumbel(game, sess, use_gumbel_noise=True, m=int(fx["m"]), c_visit=float(fx["c_visit"]), c_scale=float(fx["c_scale"]),
                       seed=int(fx["seed"]), lib_path=emu_lib)
    for ply in range(12):
        move, rows = mcts.run(iteration_limit=int(fx["run_iterations"]), use_bar=False)
        assert int(move) == fx["actions"][ply]
        pol = np.zeros(7, np.float32)
        for r in rows:
            pol[int(r[0])] = r[1]
        np.testing.assert_array_equal(pol, fx["policies"][ply])
        game.do_action(move)
        mcts.prune_tree(move)
    mcts.close()


def test_mcts_run_with_time_limit(emu_lib, oracle):
    """MCTS.run(time_limit=...) (MCTS.py:528-563): the search stops when the host clock says so, the rows stay consistent, and an
    iteration limit given together with it still caps the search."""
    import time
    from grok_alpha_zero_amd.mcts import MCTS
    game = GAMES["Connect4"]()
    mcts = MCTS(game, None, c_puct_init=2.5, dirichlet_alpha=0.5, tau=0.0, seed=3, hash_salt=1, lib_path=emu_lib)
    t0 = time.time()
    move, rows = mcts.run(iteration_limit=None, time_limit=0.2, use_bar=False)
    dt = time.time() - t0
    assert 0.15 <= dt < 5.0
    visits = sum(int(r[4]) for r in rows)
    assert visits >= 7 and rows == sorted(rows, key=lambda r: r[4], reverse=True)
    assert int(move) == int(rows[0][0])                                       # tau = 0: the most visited move
    game.do_action(move); mcts.prune_tree(move)
    move2, rows2 = mcts.run(iteration_limit=40, time_limit=60.0, use_bar=False)   # the iteration limit ends it long before the clock
    # the kept subtree carries its visits (MCTS.py:654): at most 40 new simulations on top of it
    assert sum(int(r[4]) for r in rows2) <= 40 + visits
    mcts.close()


def _drive_puct_fixture(MCTS, fx, lib_path, oracle, with_session):
    import json
    game = GAMES[str(fx["game"])]()
    A = game.policy_shape[0]
    sess = _HashSession(oracle, A, int(fx["salt"])) if with_session else None
    mcts = MCTS(game, sess, c_puct_init=float(fx["c_puct_init"]), use_dirichlet=True, dirichlet_alpha=float(fx["dirichlet_alpha"]),
                dirichlet_epsilon=0.25, tau=1.0, seed=int(fx["seed"]), hash_salt=int(fx["salt"]), lib_path=lib_path)
    taus = fx["taus"].tolist()
    updates = {int(k): v for k, v in json.loads(str(fx["update_json"])).items()}
    for ply in range(len(fx["actions"])):
        mcts.update_hyperparams(tau=(1.0 if ply < 4 else 0) if not taus else taus[ply % len(taus)])
        if ply in updates:
            mcts.update_hyperparams(**updates[ply])
        move, rows = mcts.run(iteration_limit=int(fx["iteration_limit"]), use_bar=False)
        N = np.zeros(A, np.uint32); Wv = np.zeros(A, np.float32); P = np.zeros(A, np.float32); Tm = np.full(A, -9, np.int32)
        for r in rows:
            a = type(game).action_to_index(r[0]); N[a] = r[4]; Wv[a] = r[3]; P[a] = r[5]; Tm[a] = -2 if r[7] is None else int(r[7])
        np.testing.assert_array_equal(N, fx["root_N"][ply], err_msg=f"N ply {ply}")
        np.testing.assert_array_equal(Wv, fx["root_W"][ply], err_msg=f"W ply {ply}")
        np.testing.assert_array_equal(P, fx["root_P"][ply], err_msg=f"P ply {ply}")
        np.testing.assert_array_equal(Tm, fx["is_terminal"][ply], err_msg=f"is_terminal ply {ply}")   # rows[..][7] = child.is_terminal
        assert type(game).action_to_index(move) == fx["actions"][ply], f"sampled move, ply {ply} (tau {mcts.tau})"
        game.do_action(move)
        if game.check_win() != -2:
            break
        mcts.prune_tree(move)
    assert ply == len(fx["actions"]) - 1
    if with_session:
        assert sess.calls == int(fx["evaluator_calls"])
    mcts.close()


@pytest.mark.parametrize("name", ["c4_mcts_single_tau", "ttt_mcts_single_tau", "c4_mcts_single_update"])
def test_mcts_class_general_tau_and_hyperparam_updates(emu_lib, oracle, name):
    """tau outside {0, 1}: weights N^(1/tau) in float64 (MCTS.py:606-610), and update_hyperparams(c_puct_init / c_puct_base /
    dirichlet_alpha / dirichlet_epsilon) between moves (MCTS.py:134-168), against fixtures recorded from the reference's class."""
    from grok_alpha_zero_amd.mcts import MCTS
    _drive_puct_fixture(MCTS, np.load(os.path.join(GOLDEN, name + ".npz")), emu_lib, oracle, with_session=name.startswith("ttt"))


def _drive_gumbel_fixture(MCTS_Gumbel, fx, lib_path, oracle):
    import json
    game = GAMES[str(fx["game"])]()
    A = game.policy_shape[0]
    sess = _HashSession(oracle, A, int(fx["salt"]))
    mcts = MCTS_Gumbel(game, sess, use_gumbel_noise=bool(int(fx["use_gumbel_noise"])), m=int(fx["m"]), c_visit=float(fx["c_visit"]),
                       c_scale=float(fx["c_scale"]), seed=int(fx["seed"]), lib_path=lib_path)
    updates = {int(k): v for k, v in json.loads(str(fx["update_json"])).items()}
    for ply in range(len(fx["actions"])):
        if ply in updates:
            mcts.update_hyperparams(**updates[ply])
        move, rows = mcts.run(iteration_limit=int(fx["iteration_limit"]), use_bar=False)
        pol = np.zeros(A, np.float32); N = np.zeros(A, np.uint32); Wv = np.zeros(A, np.float32); P = np.zeros(A, np.float32)
        for r in rows:
            a = type(game).action_to_index(r[0]); pol[a] = r[1]; N[a] = r[4]; Wv[a] = r[3]; P[a] = r[5]
        for k, v in (("root_N", N), ("root_W", Wv), ("root_P", P), ("policies", pol)):
            np.testing.assert_array_equal(v, fx[k][ply], err_msg=f"{k} ply {ply}")
        assert type(game).action_to_index(move) == fx["actions"][ply]
        game.do_action(move)
        if game.check_win() != -2:
            break
        mcts.prune_tree(move)
    assert ply == len(fx["actions"]) - 1 and sess.calls == int(fx["evaluator_calls"])
    mcts.close()


@pytest.mark.parametrize("name", ["c4_gsingle_nonoise", "ttt_gsingle_nonoise", "c4_gsingle_update"])
def test_mcts_gumbel_class_without_noise_and_with_updates(emu_lib, oracle, name):
    """MCTS_Gumbel(use_gumbel_noise=False) — the class default (MCTS_Gumbel.py:157,592-596) — and update_hyperparams(m, c_visit,
    c_scale) (MCTS_Gumbel.py:186-210), against fixtures recorded from the reference's class."""
    from grok_alpha_zero_amd.mcts import MCTS_Gumbel
    _drive_gumbel_fixture(MCTS_Gumbel, np.load(os.path.join(GOLDEN, name + ".npz")), emu_lib, oracle)
