"""CPU suite: the oracle (oracle/*.c) against the golden vectors generated from the reference itself
(tools/gen_golden.py, reference imported under tools/ref_shim.py in the build container).

Bar: integer visit counts N, root.visits, chosen actions, input states bit-exact; W, P, policies, values
bit-exact too (float32 results of the same operation order)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_puct_*.npz")) if "_open" not in p)


def _run(oracle, fx, use_libm):
    return oracle.selfplay_game(str(fx["game"]), int(fx["run_iterations"]), int(fx["max_actions"]),
                                int(fx["explore_first"]), int(fx["explore_second"]), float(fx["c_puct_init"]),
                                float(fx["dirichlet_alpha"]), int(fx["seed"]), int(fx["slot"]), int(fx["game_seq"]),
                                hash_salt=int(fx["salt"]), use_libm=use_libm)


@pytest.mark.parametrize("use_libm", [False, True], ids=["detmath", "libm"])
@pytest.mark.parametrize("name", CASES)
def test_puct_selfplay_matches_reference(oracle, name, use_libm):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    r = _run(oracle, fx, use_libm)
    assert r["T"] == len(fx["actions"])
    np.testing.assert_array_equal(r["actions"], fx["actions"])
    np.testing.assert_array_equal(r["root_N"], fx["root_N"])          # integer visit counts: bit-exact
    np.testing.assert_array_equal(r["root_visits"], fx["root_visits"])
    np.testing.assert_array_equal(r["root_W"], fx["root_W"])          # same f32 add order -> bit-exact
    np.testing.assert_array_equal(r["root_P"], fx["root_P"])
    np.testing.assert_array_equal(r["states"], fx["states"])
    np.testing.assert_array_equal(r["policies"], fx["policies"])
    np.testing.assert_array_equal(r["values"], fx["values"].reshape(-1))
    assert r["total_evals"] == int(fx["evaluator_calls"])
    gs = fx["game_stats"]
    assert gs[0] == r["T"] and gs[1] == r["T"] and gs[2] == 1 and gs[r["winner"] + 4] == 1


def test_fixture_inventory():
    assert {"ttt_puct_a", "c4_puct_a", "c4_puct_c", "gmk_puct_a"} <= set(CASES)


GUMBEL_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_gumbel_*.npz")))


@pytest.mark.parametrize("use_libm", [False, True], ids=["detmath", "libm"])
@pytest.mark.parametrize("name", GUMBEL_CASES)
def test_gumbel_selfplay_matches_reference(oracle, name, use_libm):
    """MCTS_Gumbel (sequential halving, completed-Q, deterministic selection) + the use_gumbel branch of Self_Play."""
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    r = oracle.selfplay_game_gumbel(str(fx["game"]), int(fx["run_iterations"]), int(fx["max_actions"]), int(fx["m"]),
                                    float(fx["c_visit"]), float(fx["c_scale"]), int(fx["seed"]), int(fx["slot"]),
                                    int(fx["game_seq"]), hash_salt=int(fx["salt"]), use_libm=use_libm,
                                    stablemax=bool(int(fx["stablemax"])) if "stablemax" in fx else False)
    assert r["T"] == len(fx["actions"]) and r["total_evals"] == int(fx["evaluator_calls"])
    for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "states", "policies"):
        np.testing.assert_array_equal(r[k], fx[k], err_msg=k)
    np.testing.assert_array_equal(r["values"], fx["values"].reshape(-1))


def test_gumbel_fixture_inventory():
    assert {"ttt_gumbel_a", "c4_gumbel_a", "gmk_gumbel_a", "c4_gumbel_stable_a", "ttt_gumbel_stable_a", "gmk_gumbel_stable_a"} <= set(GUMBEL_CASES)


OPENING_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_puct_open*.npz")))


@pytest.mark.parametrize("name", OPENING_CASES)
def test_opening_actions_match_reference(oracle, name):
    """train_config["opening_actions"]: move 0 is drawn from the configured openings (Self_Play.py:130-140)."""
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    opening = list(zip(fx["opening_idx"].tolist(), fx["opening_w"].tolist()))
    r = oracle.selfplay_game(str(fx["game"]), int(fx["run_iterations"]), int(fx["max_actions"]), int(fx["explore_first"]),
                             int(fx["explore_second"]), float(fx["c_puct_init"]), float(fx["dirichlet_alpha"]), int(fx["seed"]),
                             int(fx["slot"]), int(fx["game_seq"]), hash_salt=int(fx["salt"]), opening_actions=opening)
    for k in ("actions", "root_N", "root_visits", "root_W", "states", "policies"):
        np.testing.assert_array_equal(r[k], fx[k], err_msg=k)
    np.testing.assert_array_equal(r["values"], fx["values"].reshape(-1))


def test_opening_fixtures_exercise_the_override():
    assert len(OPENING_CASES) >= 3
    differs = [bool(np.load(os.path.join(GOLDEN, n + ".npz"))["actions"][0] != np.load(os.path.join(GOLDEN, n + ".npz"))["search_actions"][0]) for n in OPENING_CASES]
    assert any(differs)        # at least one fixture actually overrode the search's move
