"""GPU suite (-m gpu): the HIP engine (libgaz_engine.so, gfx950) through the C ABI against
 (1) the golden vectors recorded from the reference and (2) the CPU oracle on the same seeded inputs.
Bar: actions, visit counts N, root.visits bit-exact; W, P, policies, values bit-exact as well (same f32/f64
operation order on both sides)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, oracle_many

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_puct_*.npz")) if "_open" not in p)


def _engine(*a, **k):
    import torch
    assert torch.cuda.is_available(), "GPU test needs a GPU"
    from grok_alpha_zero_amd.engine import SelfPlayEngine
    return SelfPlayEngine(*a, **k)


def _play_until(eng, want, max_calls=4000, waves=64):
    recs = []
    for _ in range(max_calls):
        eng.run_waves(waves)
        recs += eng.drain_finished()
        if want(recs):
            return recs
    raise AssertionError("games did not finish")


@pytest.mark.parametrize("name", CASES)
def test_hip_engine_matches_reference_fixture(name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    seq = int(fx["game_seq"])
    eng = _engine(str(fx["game"]), 1, int(fx["run_iterations"]), int(fx["max_actions"]), int(fx["explore_first"]),
                  int(fx["explore_second"]), float(fx["c_puct_init"]), float(fx["dirichlet_alpha"]), int(fx["seed"]),
                  slot_offset=int(fx["slot"]), hash_salt=int(fx["salt"]), ring_capacity=8)
    recs = _play_until(eng, lambda rs: any(r["game_seq"] == seq for r in rs))
    r = [x for x in recs if x["game_seq"] == seq][0]
    assert r["T"] == len(fx["actions"])
    for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies"):
        np.testing.assert_array_equal(r[k], fx[k], err_msg=k)
    np.testing.assert_array_equal(r["values"], fx["values"].reshape(-1))
    eng.close()


@pytest.mark.parametrize("game,G,iters,max_actions,ef,es,c,alpha", [
    ("Connect4", 256, 60, 42, 8, 7, 2.5, 0.5),
    ("TicTacToe", 128, 51, 9, 2, 1, 1.25, 1.0),
    ("Gomoku", 16, 40, 10, 6, 4, 4.5, 0.05),
])
def test_hip_engine_matches_oracle_many_games(oracle, game, G, iters, max_actions, ef, es, c, alpha):
    """Continuous device-resident self-play: every finished game (first two per slot) equals the oracle's."""
    eng = _engine(game, G, iters, max_actions, ef, es, c, alpha, seed=2024, hash_salt=17, slot_offset=1000,
                  ring_capacity=4 * G)
    recs = _play_until(eng, lambda rs: len({(r["slot"], r["game_seq"]) for r in rs if r["game_seq"] < 2}) == 2 * G)
    checked = 0
    recs = [r for r in recs if r["game_seq"] < 2]
    ora = oracle_many(oracle.selfplay_game, [((game, iters, max_actions, ef, es, c, alpha, 2024, r["slot"], r["game_seq"]), dict(hash_salt=17)) for r in recs])
    for r, o in zip(recs, ora):
        assert r["T"] == o["T"] and r["winner"] == o["winner"]
        for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values", "q"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']} seq {r['game_seq']}")
        checked += 1
    assert checked == 2 * G
    st = eng.stats()
    assert st["game_stats"][2] >= 2 * G and st["game_stats"][3] + st["game_stats"][4] + st["game_stats"][5] == st["game_stats"][2]
    eng.close()


def test_hip_sync_api_and_host_moves(oracle):
    """run_move / get_root_stats / apply_moves; host-chosen moves (opening_actions style override)."""
    from grok_alpha_zero_amd.engine import PH_HALT
    G, iters = 32, 40
    eng = _engine("Connect4", G, iters, 42, 8, 7, 2.5, 0.5, seed=9, hash_salt=1, sync_moves=True)
    ora = [oracle.selfplay_game("Connect4", iters, 42, 8, 7, 2.5, 0.5, 9, g, 0, hash_salt=1) for g in range(G)]
    for ply in range(42):
        eng.run_move()
        st = eng.root_stats()
        live = [g for g in range(G) if st["phase"][g] != PH_HALT]
        if not live:
            break
        for g in live:
            np.testing.assert_array_equal(st["N"][g], ora[g]["root_N"][ply])
            assert st["chosen"][g] == ora[g]["actions"][ply]
        eng.apply_moves(st["chosen"])      # explicit host moves == sampled moves
    recs = {r["slot"]: r for r in eng.drain_finished()}
    assert len(recs) == G and all(recs[g]["T"] == ora[g]["T"] for g in range(G))
    eng.close()


def _legal_count(game, actions, ply):
    if game == "Connect4":
        return int((np.bincount(actions[:ply], minlength=7) < 6).sum())
    return 225 - ply


def _check_puct_invariants(game, r, run_iterations, A, max_actions):
    """Size-independent properties of one finished PUCT self-play game (MCTS.py:528-618, Self_Play.py:71-157)."""
    T = r["T"]
    assert 1 <= T <= max_actions and len(r["actions"]) == T
    np.testing.assert_allclose(r["policies"].sum(1), 1.0, atol=1e-6)
    n_sum = r["root_N"].sum(1).astype(np.int64)
    rv = r["root_visits"].astype(np.int64)
    # every simulation of run() adds >= 1 visit to the root and to exactly one root edge.  A root made by create_expand_root starts
    # at 0 visits (sum N == visits); a re-rooted node carries the visit that expanded it (sum N == visits - 1) unless it is a
    # terminal parent (its creation backed one visit per terminal child up: sum N == visits).
    assert ((n_sum == rv) | (n_sum == rv - 1)).all(), (n_sum, rv)
    for ply in range(T):
        legal = _legal_count(game, r["actions"], ply)
        lim = 1 if legal == 1 else (run_iterations if run_iterations >= legal else 3 * legal)          # MCTS.py:542-548
        assert rv[ply] >= lim, (ply, rv[ply], lim)                # the iteration budget was spent (carried visits come on top)
        assert (r["root_N"][ply] > 0).sum() <= legal and r["root_N"][ply][r["actions"][ply]] > 0     # the played move was searched
    # both trees are fresh for their first move (tree 2's empty root cannot be re-rooted, MCTS.py:661-671): no carried visits, so
    # sum N == root.visits exactly; visits exceed the budget only by what terminal parents deep in the tree backed up (MCTS.py:373-380)
    for ply in range(min(2, T)):
        assert n_sum[ply] == rv[ply] and run_iterations <= rv[ply] <= run_iterations + 8 * A, (ply, rv[ply], n_sum[ply])
    assert np.all(np.abs(r["values"]) <= 1.0) and np.all(np.abs(r["q"]) <= 1.0)
    z = r["z"]
    assert r["winner"] in (-1, 0, 1) and (np.all(z == 0) if r["winner"] == 0 else np.all(np.abs(z) == 1))


def _first_games(eng, G, want_slots, n_records, max_calls=4000, waves=128):
    recs = []
    for _ in range(max_calls):
        eng.run_waves(waves)
        recs += eng.drain_finished()
        have = {r["slot"] for r in recs if r["game_seq"] == 0}
        if len(recs) >= n_records and want_slots <= have:
            return recs
    raise AssertionError(f"only {len(recs)} games finished")


def test_full_size_connect4_parity_and_invariants(oracle):
    """BASELINE configs[1] at its own size — 4096 concurrent Connect4 games, 200 sims/move (synthetic evaluator): the first game of
    64 slots spread over the batch equals the oracle's game bit for bit, and EVERY finished game satisfies the visit-count
    bookkeeping of MCTS.run (budget spent, sum N vs root.visits, policies normalised, legal games)."""
    G = 4096
    eng = _engine("Connect4", G, 200, 42, 8, 7, 2.5, 0.5, seed=1, hash_salt=3, ring_capacity=2 * G)
    slots = set(range(0, G, 64))
    recs = _first_games(eng, G, slots, 1024, max_calls=600)
    for r in recs:
        _check_puct_invariants("Connect4", r, 200, 7, 42)
        assert np.bincount(r["actions"], minlength=7).max() <= 6
    for r in recs:
        if r["game_seq"] == 0 and r["slot"] in slots:
            o = oracle.selfplay_game("Connect4", 200, 42, 8, 7, 2.5, 0.5, 1, r["slot"], 0, hash_salt=3)
            assert r["T"] == o["T"] and r["winner"] == o["winner"]
            for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values", "q"):
                np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']}")
    st = eng.stats()["game_stats"]
    assert st[2] >= 1024 and st[3] + st[4] + st[5] == st[2] and st[0] <= 42
    eng.close()


def test_gomoku_config_parameters_natural_ends_vs_oracle(oracle):
    """BASELINE configs[3] parameters — Gomoku 15x15, 400 sims/move, c_puct 4.5, alpha 0.05, re-root compaction on (4.2 KB node
    records) — with max_actions = the whole board so that games end by five in a row INSIDE real games: 64 concurrent games, every
    first game equal to the oracle's bit for bit (Gomoku had never been compared at depth, nor to a natural end)."""
    G, iters = 64, 400
    eng = _engine("Gomoku", G, iters, 225, 6, 4, 4.5, 0.05, seed=77, hash_salt=9, slot_offset=300, ring_capacity=4 * G)
    recs = _first_games(eng, G, set(range(300, 300 + G)), G, max_calls=20000, waves=256)
    natural = 0
    recs = [r for r in recs if r["game_seq"] == 0]
    ora = oracle_many(oracle.selfplay_game, [(("Gomoku", iters, 225, 6, 4, 4.5, 0.05, 77, r["slot"], 0), dict(hash_salt=9)) for r in recs], threads=12)
    for r, o in zip(recs, ora):
        assert r["T"] == o["T"] and r["winner"] == o["winner"]
        for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values", "q"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']}")
        _check_puct_invariants("Gomoku", r, iters, 225, 225)
        natural += r["winner"] in (-1, 1) and r["T"] < 225
    assert natural >= G // 2, f"only {natural} of {G} games ended by five in a row"
    eng.close()


def test_full_size_gomoku_parity_and_invariants(oracle):
    """BASELINE configs[3] at its own size: 2048 concurrent Gomoku games, 400 sims/move, max_actions 150 (bench.py's value), arena
    with re-root compaction (80 GB of tree records).  First game of 8 slots vs the oracle + invariants on every finished game."""
    G, iters, max_actions = 2048, 400, 150
    eng = _engine("Gomoku", G, iters, max_actions, 6, 4, 4.5, 0.05, seed=5, hash_salt=11, ring_capacity=G)
    slots = set(range(0, G, 256))
    recs = _first_games(eng, G, slots, 96, max_calls=3000, waves=256)
    for r in recs:
        _check_puct_invariants("Gomoku", r, iters, 225, max_actions)
        assert len(set(r["actions"].tolist())) == r["T"]                  # no cell played twice
    cmp = [r for r in recs if r["game_seq"] == 0 and r["slot"] in slots]
    ora = oracle_many(oracle.selfplay_game, [(("Gomoku", iters, max_actions, 6, 4, 4.5, 0.05, 5, r["slot"], 0), dict(hash_salt=11)) for r in cmp])
    for r, o in zip(cmp, ora):
        if True:
            assert r["T"] == o["T"] and r["winner"] == o["winner"]
            for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values"):
                np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']}")
    eng.close()


def test_full_size_gumbel_parity_and_invariants(oracle):
    """BASELINE configs[4] at its own size: 8192 concurrent Connect4 games, Gumbel search n = 32, m = 7.  First game of 64 slots vs
    the oracle; on every finished game: the improved policy is a distribution over legal moves that supports the played move, and the
    root's visit bookkeeping is consistent (sum N == root.visits: the tree is rebuilt every move, Self_Play.py:151-153)."""
    from grok_alpha_zero_amd.engine import SEARCH_GUMBEL
    G, n, m = 8192, 32, 7
    eng = _engine("Connect4", G, n, 42, 0, 0, 0.0, 0.0, seed=12, hash_salt=4, ring_capacity=2 * G, search=SEARCH_GUMBEL, gumbel_m=m,
                  c_visit=50.0, c_scale=1.0)
    slots = set(range(0, G, 128))
    recs = _first_games(eng, G, slots, 2048, max_calls=600)
    for r in recs:
        T = r["T"]
        assert 7 <= T <= 42 and np.bincount(r["actions"], minlength=7).max() <= 6
        np.testing.assert_allclose(r["policies"].sum(1), 1.0, atol=1e-5)
        np.testing.assert_array_equal(r["root_N"].sum(1).astype(np.int64), r["root_visits"].astype(np.int64))   # a fresh tree every move
        for ply in range(T):
            full = np.bincount(r["actions"][:ply], minlength=7) >= 6
            assert (r["policies"][ply][full] == 0).all() and (r["root_N"][ply][full] == 0).all()    # nothing on full columns
            assert r["policies"][ply][r["actions"][ply]] > 0                                          # the played move has support
            assert r["root_visits"][ply] <= n * 7 + m           # n simulations (a terminal parent backs up to 7 visits) + m expansions
    for r in recs:
        if r["game_seq"] == 0 and r["slot"] in slots:
            o = oracle.selfplay_game_gumbel("Connect4", n, 42, m, 50.0, 1.0, 12, r["slot"], 0, hash_salt=4)
            assert r["T"] == o["T"] and r["winner"] == o["winner"]
            for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values", "q"):
                np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']}")
            # the opening move's budget: n simulations on top of the m root-child expansions (root.visits = 38)
            assert o["root_visits"][0] >= n
    eng.close()


GUMBEL_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_gumbel_*.npz")))


@pytest.mark.parametrize("name", GUMBEL_CASES)
def test_hip_gumbel_engine_matches_reference_fixture(name):
    from grok_alpha_zero_amd.engine import SEARCH_GUMBEL
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    seq = int(fx["game_seq"])
    eng = _engine(str(fx["game"]), 1, int(fx["run_iterations"]), int(fx["max_actions"]), 0, 0, 0.0, 0.0, int(fx["seed"]),
                  slot_offset=int(fx["slot"]), hash_salt=int(fx["salt"]), ring_capacity=8, search=SEARCH_GUMBEL,
                  gumbel_m=int(fx["m"]), c_visit=float(fx["c_visit"]), c_scale=float(fx["c_scale"]),
                  gumbel_stablemax=bool(int(fx["stablemax"])) if "stablemax" in fx else False)
    recs = _play_until(eng, lambda rs: any(r["game_seq"] == seq for r in rs))
    r = [x for x in recs if x["game_seq"] == seq][0]
    assert r["T"] == len(fx["actions"])
    for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies"):
        np.testing.assert_array_equal(r[k], fx[k], err_msg=k)
    np.testing.assert_array_equal(r["values"], fx["values"].reshape(-1))
    eng.close()


@pytest.mark.parametrize("game,G,n,max_actions,m", [("Connect4", 512, 32, 42, 7), ("TicTacToe", 64, 16, 9, 4), ("Gomoku", 8, 48, 8, 16)])
def test_hip_gumbel_matches_oracle_many_games(oracle, game, G, n, max_actions, m):
    """BASELINE config 5 shape (Connect4, n = 32, m = 7) on many concurrent games vs the oracle, bit-exact."""
    from grok_alpha_zero_amd.engine import SEARCH_GUMBEL
    eng = _engine(game, G, n, max_actions, 0, 0, 0.0, 0.0, seed=31, hash_salt=13, slot_offset=500, ring_capacity=4 * G,
                  search=SEARCH_GUMBEL, gumbel_m=m, c_visit=50.0, c_scale=1.0)
    recs = _play_until(eng, lambda rs: len({r["slot"] for r in rs if r["game_seq"] == 0}) == G)
    checked = 0
    for r in recs:
        if r["game_seq"] != 0:
            continue
        o = oracle.selfplay_game_gumbel(game, n, max_actions, m, 50.0, 1.0, 31, r["slot"], 0, hash_salt=13)
        assert r["T"] == o["T"] and r["winner"] == o["winner"]
        for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values", "q"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']}")
        checked += 1
    assert checked == G
    eng.close()


@pytest.mark.parametrize("game,G,iters,kw", [
    ("Connect4", 192, 48, dict(create_new_root=True)),
    ("Connect4", 192, 48, dict(c_puct_base=300.0)),
    ("Connect4", 128, 40, dict(opening_actions=[(3, 0.5), (2, 0.25), (4, 0.25)])),
    ("TicTacToe", 96, 30, dict(create_new_root=True, c_puct_base=50.0)),
])
def test_hip_puct_options_match_oracle_many_games(oracle, game, G, iters, kw):
    """The PUCT self-play options at scale — create_new_root (no tree reuse, Self_Play.py:147-157), a non-default c_puct_base
    (MCTS.py:172-191) and train_config["opening_actions"] (Self_Play.py:130-140) — on many concurrent games vs the oracle, bit-exact."""
    ma, ef, es, c, alpha = (42, 8, 7, 2.5, 0.5) if game == "Connect4" else (9, 2, 1, 1.25, 1.0)
    eng = _engine(game, G, iters, ma, ef, es, c, alpha, seed=99, hash_salt=21, slot_offset=7, ring_capacity=4 * G, **kw)
    recs = _play_until(eng, lambda rs: len({(r["slot"], r["game_seq"]) for r in rs if r["game_seq"] < 2}) == 2 * G)
    recs = [r for r in recs if r["game_seq"] < 2]
    ora = oracle_many(oracle.selfplay_game, [((game, iters, ma, ef, es, c, alpha, 99, r["slot"], r["game_seq"]), dict(hash_salt=21, **kw)) for r in recs])
    assert len(recs) == 2 * G
    for r, o in zip(recs, ora):
        assert r["T"] == o["T"] and r["winner"] == o["winner"]
        for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values", "q"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']} seq {r['game_seq']}")
    eng.close()


@pytest.mark.parametrize("G,n,m,stablemax,noise", [(256, 32, 7, True, True), (192, 24, 5, False, False), (96, 40, 7, True, False)])
def test_hip_gumbel_variants_match_oracle_many_games(oracle, G, n, m, stablemax, noise):
    """The Gumbel engine's two switches at scale — activation_fn = "stablemax" in the completed-Q selection (MCTS_Gumbel.py:77-99,
    Self_Play.py:69) and use_gumbel_noise = False (MCTS_Gumbel.py:157,592) — on many concurrent Connect4 games vs the oracle, bit-exact
    (the reference fixtures pin each switch on single games)."""
    from grok_alpha_zero_amd.engine import SEARCH_GUMBEL
    eng = _engine("Connect4", G, n, 42, 0, 0, 0.0, 0.0, seed=77, hash_salt=5, slot_offset=40, ring_capacity=4 * G, search=SEARCH_GUMBEL,
                  gumbel_m=m, c_visit=50.0, c_scale=1.0, gumbel_stablemax=stablemax, use_gumbel_noise=noise)
    recs = _play_until(eng, lambda rs: len({r["slot"] for r in rs if r["game_seq"] == 0}) == G)
    first = [r for r in recs if r["game_seq"] == 0]
    ora = oracle_many(oracle.selfplay_game_gumbel, [(("Connect4", n, 42, m, 50.0, 1.0, 77, r["slot"], 0), dict(hash_salt=5, stablemax=stablemax, gumbel_noise=noise))
                                                      for r in first])
    assert len(first) == G
    for r, o in zip(first, ora):
        assert r["T"] == o["T"] and r["winner"] == o["winner"]
        for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values", "q"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']}")
    eng.close()


def test_hip_reroot_compaction_matches_oracle(oracle):
    """Re-root compaction forced on for Connect4 (it is the default only for Gomoku): results must not change."""
    G, iters = 128, 50
    eng = _engine("Connect4", G, iters, 42, 8, 7, 2.5, 0.5, seed=3, hash_salt=2, ring_capacity=4 * G, compact_trees=1)
    recs = _play_until(eng, lambda rs: len({r["slot"] for r in rs if r["game_seq"] == 0}) == G)
    for r in recs:
        if r["game_seq"] == 0:
            o = oracle.selfplay_game("Connect4", iters, 42, 8, 7, 2.5, 0.5, 3, r["slot"], 0, hash_salt=2)
            for k in ("actions", "root_N", "root_visits", "root_W", "policies"):
                np.testing.assert_array_equal(r[k], o[k], err_msg=k)
    eng.close()


@pytest.mark.parametrize("name", ["c4_mcts_single", "c4_mcts_single_ffw"])
def test_hip_mcts_class_matches_reference_class(oracle, name):
    """grok_alpha_zero_amd.mcts.MCTS on the GPU vs the fixture recorded from the reference's MCTS class (single tree; _ffw:
    constructed with fast_find_win=True)."""
    from grok_alpha_zero_amd.games import GAMES
    from grok_alpha_zero_amd.mcts import MCTS
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    game = GAMES["Connect4"]()
    mcts = MCTS(game, None, c_puct_init=float(fx["c_puct_init"]), dirichlet_alpha=float(fx["dirichlet_alpha"]), tau=1.0,
                fast_find_win=bool(int(fx["fast_find_win"])) if "fast_find_win" in fx else False,
                seed=int(fx["seed"]), hash_salt=int(fx["salt"]))
    for ply in range(len(fx["actions"])):
        mcts.update_hyperparams(tau=1.0 if ply < 4 else 0)
        move, rows = mcts.run(iteration_limit=int(fx["iteration_limit"]), use_bar=False)
        N = np.zeros(7, np.uint32)
        for r in rows:
            N[int(r[0])] = r[4]
        assert int(move) == fx["actions"][ply]
        np.testing.assert_array_equal(N, fx["root_N"][ply])
        game.do_action(move)
        if game.check_win() != -2:
            break
        mcts.prune_tree(move)
    mcts.close()


@pytest.mark.parametrize("name", ["c4_mcts_single_tau", "c4_mcts_single_update"])
def test_hip_mcts_class_general_tau_and_updates(oracle, name):
    """tau outside {0, 1} (weights N^(1/tau), MCTS.py:606-610) and update_hyperparams between moves (MCTS.py:134-168) on the GPU,
    against fixtures recorded from the reference's MCTS class."""
    from test_mcts_classes import _drive_puct_fixture
    from grok_alpha_zero_amd.mcts import MCTS
    _drive_puct_fixture(MCTS, np.load(os.path.join(GOLDEN, name + ".npz")), None, oracle, with_session=False)


@pytest.mark.parametrize("name", ["c4_gsingle_nonoise", "c4_gsingle_update"])
def test_hip_mcts_gumbel_class_without_noise_and_with_updates(oracle, name):
    """MCTS_Gumbel(use_gumbel_noise=False) (the class default, MCTS_Gumbel.py:157) and update_hyperparams(m, c_visit, c_scale) on the
    GPU, one session.run per simulation through GAZ_EVAL_EXTERNAL, against fixtures recorded from the reference's class."""
    from test_mcts_classes import _drive_gumbel_fixture
    from grok_alpha_zero_amd.mcts import MCTS_Gumbel
    _drive_gumbel_fixture(MCTS_Gumbel, np.load(os.path.join(GOLDEN, name + ".npz")), None, oracle)


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_puct_open*.npz"))))
def test_hip_opening_actions_match_reference(name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    eng = _engine(str(fx["game"]), 1, int(fx["run_iterations"]), int(fx["max_actions"]), int(fx["explore_first"]),
                  int(fx["explore_second"]), float(fx["c_puct_init"]), float(fx["dirichlet_alpha"]), int(fx["seed"]),
                  slot_offset=int(fx["slot"]), hash_salt=int(fx["salt"]), ring_capacity=8,
                  opening_actions=list(zip(fx["opening_idx"].tolist(), fx["opening_w"].tolist())))
    r = _play_until(eng, lambda rs: len(rs) >= 1)[0]
    for k in ("actions", "root_N", "root_visits", "root_W", "policies"):
        np.testing.assert_array_equal(r[k], fx[k], err_msg=k)
    eng.close()


@pytest.mark.parametrize("G,ring", [(512, 8192), (3200, 16384)])
def test_hip_evaluation_cache_gives_identical_games(G, ring):
    """eval_cache_log2 (on-device evaluation cache): with the HIP ResNet evaluator the finished games are bit-identical with the
    cache on and off (rows of a batch are independent, so cached outputs equal fresh ones), and a good share of requests hit.
    3200 games: the evaluator launch then mixes both tile shapes of k_trunk_mix (1024 three-board + 64 two-board workgroups), and a
    position cached out of one shape is later compared with fresh outputs out of the other."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
    from grok_alpha_zero_amd.net import Connect4Net
    w = Connect4Net(2, seed=3).eval().export_engine_weights()
    got = []
    for log2 in (0, 16):
        eng = SelfPlayEngine("Connect4", G, 32, 16, 4, 3, 2.5, 0.5, seed=21, evaluator=EVAL_RESNET, net_blocks=2, ring_capacity=ring,
                             eval_cache_log2=log2, game_groups=1)
        eng.load_weights(w)
        eng.run_waves(900); eng.synchronize()
        st = eng.stats()
        got.append(({(r["slot"], r["game_seq"]): r for r in eng.drain_finished()}, st["cache_hits"], st["evals"]))
        eng.close()
    (ra, h0, e0), (rb, h1, e1) = got
    assert h0 == 0 and h1 > 0.1 * e1
    common = sorted(set(ra) & set(rb))
    assert len(common) >= G and len(rb) >= len(ra)
    for k in common:
        for f in ("actions", "root_N", "root_W", "root_P", "policies", "q", "evals", "root_visits", "winner", "T"):
            np.testing.assert_array_equal(np.asarray(ra[k][f]), np.asarray(rb[k][f]), err_msg=f"{k} {f}")


def test_hip_repack_keeps_every_game(oracle):
    """gaz_engine_repack on the GPU (tail of a generation: the live games are moved to the lowest slots, launches shrink to them):
    every admitted game of a 1300-game generation on 512 slots still equals the oracle's."""
    from test_engine_emu import _repack_run
    ora = {}

    def fn(s, q):
        return oracle.selfplay_game("Connect4", 30, 42, 4, 3, 2.5, 0.5, 13, s, q, hash_salt=6)
    _repack_run(None, oracle, "Connect4", 512, 1300, 30, 42, {}, fn)


def test_hip_repack_with_network_fused_launch_and_cache():
    """The same with everything on that run_self_play uses: ResNet evaluator, fused tree + trunk launch, evaluation cache.  A run that
    repacks whenever half of its slots idle must produce the very records of a run that never does."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
    from grok_alpha_zero_amd.net import Connect4Net
    w = Connect4Net(2, seed=8).eval().export_engine_weights()
    G, budget = 768, 1800
    out = []
    for do_repack in (True, False):
        eng = SelfPlayEngine("Connect4", G, 24, 42, 4, 3, 2.5, 0.5, seed=3, evaluator=EVAL_RESNET, net_blocks=2, ring_capacity=4 * G,
                             eval_cache_log2=16, games_budget=budget)
        eng.load_weights(w)
        recs, launch, n_repack = [], G, 0
        for _ in range(4000):
            eng.run_waves(32)
            recs += eng.drain_finished()
            remaining = budget - len(recs)
            if remaining == 0:
                break
            if do_repack and remaining * 2 <= launch and launch > 16:
                _, launch = eng.repack(); n_repack += 1
        assert len(recs) == budget and (n_repack >= 3) == do_repack
        assert eng.stats()["fused_wave"] == 1
        eng.close()
        out.append({(r["slot"], r["game_seq"]): r for r in recs})
    a, b = out
    assert set(a) == set(b)
    for k in a:
        for f in ("actions", "root_N", "root_W", "root_P", "policies", "q", "evals", "root_visits", "winner", "T"):
            np.testing.assert_array_equal(np.asarray(a[k][f]), np.asarray(b[k][f]), err_msg=f"{k} {f}")
