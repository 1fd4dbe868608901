"""CPU suite: the engine's device code compiled for the host with a one-lane wave (tests/emu) + the real host
logic (C ABI, state machine, record ring) against the golden vectors recorded from the reference.
This is test infrastructure: the product library is the hipcc build and is covered by the -m gpu tests."""
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU = os.path.join(EMU_DIR, "libgaz_emu.so")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_puct_*.npz")) if "_open" not in p)


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return EMU


def play_fixture(fx, lib_path, **kw):
    from grok_alpha_zero_amd.engine import SelfPlayEngine
    eng = SelfPlayEngine(str(fx["game"]), 1, int(fx["run_iterations"]), int(fx["max_actions"]), int(fx["explore_first"]),
                         int(fx["explore_second"]), float(fx["c_puct_init"]), float(fx["dirichlet_alpha"]), int(fx["seed"]),
                         slot_offset=int(fx["slot"]), hash_salt=int(fx["salt"]), ring_capacity=8, lib_path=lib_path, **kw)
    recs = []
    for _ in range(20000):
        eng.run_waves(64)
        recs += eng.drain_finished()
        if any(r["game_seq"] == int(fx["game_seq"]) for r in recs):
            break
    eng.close()
    return [r for r in recs if r["game_seq"] == int(fx["game_seq"])][0]


def assert_matches_fixture(r, fx):
    assert r["T"] == len(fx["actions"])
    for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies"):
        np.testing.assert_array_equal(r[k], fx[k], err_msg=k)
    np.testing.assert_array_equal(r["values"], fx["values"].reshape(-1))
    gs = fx["game_stats"]
    assert gs[r["winner"] + 4] == 1 and gs[1] == r["T"]


@pytest.mark.parametrize("name", [c for c in CASES if not c.startswith("gmk")] + ["gmk_puct_a"])
def test_emu_engine_matches_reference_fixture(emu_lib, name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert_matches_fixture(play_fixture(fx, emu_lib), fx)


def test_emu_sync_api_matches_oracle(emu_lib, oracle):
    """run_move / get_root_stats / apply_moves (the per-move API) on 3 games."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, PH_HALT
    G, iters = 3, 30
    eng = SelfPlayEngine("Connect4", G, iters, 42, 8, 7, 2.5, 0.5, seed=5, hash_salt=9, sync_moves=True, lib_path=emu_lib)
    ora = [oracle.selfplay_game("Connect4", iters, 42, 8, 7, 2.5, 0.5, 5, g, 0, hash_salt=9) for g in range(G)]
    for ply in range(42):
        eng.run_move()
        st = eng.root_stats()
        live = [g for g in range(G) if st["phase"][g] != PH_HALT]
        if not live:
            break
        for g in live:
            assert ply < ora[g]["T"]
            np.testing.assert_array_equal(st["N"][g], ora[g]["root_N"][ply])
            np.testing.assert_array_equal(st["W"][g], ora[g]["root_W"][ply])
            assert st["chosen"][g] == ora[g]["actions"][ply] and st["root_visits"][g] == ora[g]["root_visits"][ply]
        eng.apply_moves()
    recs = {r["slot"]: r for r in eng.drain_finished()}
    assert set(recs) == set(range(G))
    for g in range(G):
        assert recs[g]["winner"] == ora[g]["winner"] and recs[g]["T"] == ora[g]["T"]
    eng.close()


def test_emu_external_evaluator_roundtrip(emu_lib, oracle):
    """wave_begin / read_batch / write_outputs: the session.run boundary with a host-side evaluator."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_EXTERNAL
    G, iters = 2, 20
    eng = SelfPlayEngine("TicTacToe", G, iters, 9, 2, 1, 1.25, 1.0, seed=3, evaluator=EVAL_EXTERNAL, lib_path=emu_lib)
    recs = []
    for _ in range(3000):
        eng.wave_begin()
        x, pend = eng.read_batch()
        pol = np.zeros((G, 9), np.float32); val = np.zeros(G, np.float32)
        for g in range(G):
            if pend[g]:
                pol[g], val[g] = oracle.hash_eval(x[g], 9, 21)
        eng.write_outputs(pol, val)
        recs += eng.drain_finished()
        if len({r["slot"] for r in recs if r["game_seq"] == 0}) == G:
            break
    for r in recs:
        if r["game_seq"] == 0:
            o = oracle.selfplay_game("TicTacToe", iters, 9, 2, 1, 1.25, 1.0, 3, r["slot"], 0, hash_salt=21)
            np.testing.assert_array_equal(r["root_N"], o["root_N"])
            np.testing.assert_array_equal(r["actions"], o["actions"])
    eng.close()


def test_c_abi_exports_every_declared_symbol():
    """The product library (hipcc build) loads and exports every entry point include/gaz_engine.h declares."""
    import re
    from grok_alpha_zero_amd import build as B, engine
    B.build_engine()
    lib = engine.load_library()
    hdr = open(os.path.join(ROOT, "include", "gaz_engine.h")).read()
    names = set(re.findall(r"\b(gaz_engine_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n


GUMBEL_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_gumbel_*.npz")))


def play_gumbel_fixture(fx, lib_path):
    from grok_alpha_zero_amd.engine import SelfPlayEngine, SEARCH_GUMBEL
    eng = SelfPlayEngine(str(fx["game"]), 1, int(fx["run_iterations"]), int(fx["max_actions"]), 0, 0, 0.0, 0.0, int(fx["seed"]),
                         slot_offset=int(fx["slot"]), hash_salt=int(fx["salt"]), ring_capacity=8, search=SEARCH_GUMBEL,
                         gumbel_m=int(fx["m"]), c_visit=float(fx["c_visit"]), c_scale=float(fx["c_scale"]),
                         gumbel_stablemax=bool(int(fx["stablemax"])) if "stablemax" in fx else False, lib_path=lib_path)
    recs = []
    for _ in range(20000):
        eng.run_waves(64)
        recs += eng.drain_finished()
        if any(r["game_seq"] == int(fx["game_seq"]) for r in recs):
            break
    eng.close()
    return [r for r in recs if r["game_seq"] == int(fx["game_seq"])][0]


@pytest.mark.parametrize("name", GUMBEL_CASES)
def test_emu_gumbel_engine_matches_reference_fixture(emu_lib, name):
    """Gumbel search (sequential halving + completed-Q selection) on the emulation build vs the reference's own output."""
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert_matches_fixture(play_gumbel_fixture(fx, emu_lib), fx)


@pytest.mark.parametrize("name", ["c4_puct_a", "c4_puct_c", "gmk_puct_a", "ttt_puct_b"])
def test_emu_reroot_compaction_does_not_change_results(emu_lib, name):
    """Double-buffered arena with breadth-first compaction at every re-root (default for Gomoku) vs the reference fixture."""
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert_matches_fixture(play_fixture(fx, emu_lib, compact_trees=1), fx)
    assert_matches_fixture(play_fixture(fx, emu_lib, compact_trees=-1), fx)


OPENING_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*_puct_open*.npz")))


@pytest.mark.parametrize("name", OPENING_CASES)
def test_emu_opening_actions_match_reference(emu_lib, name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    r = play_fixture(fx, emu_lib, opening_actions=list(zip(fx["opening_idx"].tolist(), fx["opening_w"].tolist())))
    assert_matches_fixture(r, fx)


def _finished(eng, n_waves):
    import hashlib
    eng.run_waves(n_waves); eng.synchronize()
    h = hashlib.sha256(); n = 0
    for r in sorted(eng.drain_finished(), key=lambda r: (r["slot"], r["game_seq"])):
        n += 1
        for k in ("slot", "game_seq", "winner", "T"):
            h.update(np.int64(r[k]).tobytes())
        for k in ("actions", "root_N", "root_W", "root_P", "policies", "q", "evals", "root_visits"):
            h.update(np.ascontiguousarray(r[k]).tobytes())
    return n, h.hexdigest()


@pytest.mark.parametrize("game,search,iters,max_actions", [("TicTacToe", 0, 30, 9), ("Connect4", 0, 24, 16), ("Connect4", 1, 16, 14)])
def test_evaluation_cache_changes_nothing_but_the_wave_count(game, search, iters, max_actions):
    """eval_cache_log2 > 0 (on-device evaluation cache): identical finished games — visit counts, W, priors,
    policies, q, per-move evaluator-call counts — and a non-zero hit count; fewer launches are needed for the same games."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine
    outs = []
    for log2 in (0, 12):
        eng = SelfPlayEngine(game, 24, iters, max_actions, 2, 1, 1.5, 0.8, seed=77, hash_salt=3, ring_capacity=4096, search=search,
                             gumbel_m=4 if search else 0, eval_cache_log2=log2, lib_path=EMU)
        outs.append(_finished(eng, 600) + (eng.stats()["cache_hits"], eng.stats()["evals"]))
        eng.close()
    (n0, h0, hits0, ev0), (n1, h1, hits1, ev1) = outs
    assert hits0 == 0 and hits1 > 0
    assert n0 > 24
    # the cached run finishes at least as many games in the same number of launches; compare the common prefix of games
    assert n1 >= n0
    eng_a = SelfPlayEngine(game, 24, iters, max_actions, 2, 1, 1.5, 0.8, seed=77, hash_salt=3, ring_capacity=4096, search=search,
                           gumbel_m=4 if search else 0, eval_cache_log2=0, lib_path=EMU)
    eng_b = SelfPlayEngine(game, 24, iters, max_actions, 2, 1, 1.5, 0.8, seed=77, hash_salt=3, ring_capacity=4096, search=search,
                           gumbel_m=4 if search else 0, eval_cache_log2=12, lib_path=EMU)
    eng_a.run_waves(600); eng_b.run_waves(600)
    ra = {(r["slot"], r["game_seq"]): r for r in eng_a.drain_finished()}
    rb = {(r["slot"], r["game_seq"]): r for r in eng_b.drain_finished()}
    common = sorted(set(ra) & set(rb))
    assert len(common) >= n0 // 2
    for k in common:
        for f in ("actions", "root_N", "root_W", "root_P", "policies", "q", "evals", "root_visits", "winner", "T"):
            np.testing.assert_array_equal(np.asarray(ra[k][f]), np.asarray(rb[k][f]), err_msg=f"{k} {f}")
    eng_a.close(); eng_b.close()


def _repack_run(lib_path, oracle, game, G, budget, iters, max_actions, search_kw, oracle_fn):
    """continuous self-play with a games_budget; whenever half of the covered slots have halted the live games are moved together
    (gaz_engine_repack) — every admitted game must still come out equal to the oracle's"""
    from grok_alpha_zero_amd.engine import SelfPlayEngine
    eng = SelfPlayEngine(game, G, iters, max_actions, 4, 3, 2.5, 0.5, seed=13, hash_salt=6, slot_offset=200, ring_capacity=4 * G,
                         games_budget=budget, lib_path=lib_path, **search_kw)
    recs, launch, repacks = [], G, 0
    for _ in range(100000):
        eng.run_waves(16)
        recs += eng.drain_finished()
        remaining = budget - len(recs)
        if remaining == 0:
            break
        if remaining * 2 <= launch and launch > 2:
            active, launch = eng.repack()
            assert active <= remaining and launch == max(active, 1)
            repacks += 1
    assert len(recs) == budget and repacks >= 2
    st = eng.stats()
    assert st["game_stats"][2] == budget and st["plies"] == sum(r["T"] for r in recs)      # lifetime counters survive the moves
    eng.close()
    keys = {(r["slot"], r["game_seq"]) for r in recs}
    assert keys == {(200 + g, k) for k in range(8) for g in range(G) if k * G + g < budget}
    for r in recs:
        o = oracle_fn(r["slot"], r["game_seq"])
        assert r["T"] == o["T"] and r["winner"] == o["winner"]
        for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values", "q"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=f"{k} slot {r['slot']} seq {r['game_seq']}")


def test_emu_repack_keeps_every_game(emu_lib, oracle):
    _repack_run(emu_lib, oracle, "Connect4", 24, 60, 20, 42, {},
                lambda s, q: oracle.selfplay_game("Connect4", 20, 42, 4, 3, 2.5, 0.5, 13, s, q, hash_salt=6))


def test_emu_repack_keeps_every_game_gumbel_and_compaction(emu_lib, oracle):
    from grok_alpha_zero_amd.engine import SEARCH_GUMBEL
    _repack_run(emu_lib, oracle, "Connect4", 12, 30, 16, 42, dict(search=SEARCH_GUMBEL, gumbel_m=4, c_visit=50.0, c_scale=1.0),
                lambda s, q: oracle.selfplay_game_gumbel("Connect4", 16, 42, 4, 50.0, 1.0, 13, s, q, hash_salt=6))
    _repack_run(emu_lib, oracle, "Connect4", 10, 24, 16, 42, dict(compact_trees=1),
                lambda s, q: oracle.selfplay_game("Connect4", 16, 42, 4, 3, 2.5, 0.5, 13, s, q, hash_salt=6))


@pytest.mark.parametrize("search,cache", [("puct", 0), ("gumbel", 0), ("puct", 12)])
def test_emu_skipped_evaluations_are_requested_again(emu_lib, oracle, search, cache):
    """The recovery path of the fused launch's BOUNDED hand-over (trunk.hpp TrunkArgs::spin_ticks, DevParams::eval_done): a trunk workgroup
    that gives up leaves its boards unevaluated and marks them.  gaz_engine_debug_fused_fault injects exactly that on this build (marks the
    games, poisons their outputs with NaN, raises the fault counter): the marked games must keep their requests pending and be evaluated by the
    next wave — finished games identical to the oracle's — and the host must count the faults at its next synchronisation point."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, SEARCH_GUMBEL, SEARCH_PUCT
    G, iters = 12, 24
    kw = dict(search=SEARCH_GUMBEL, gumbel_m=4, c_visit=50.0, c_scale=1.0) if search == "gumbel" else dict(search=SEARCH_PUCT)
    eng = SelfPlayEngine("Connect4", G, iters, 42, 8, 7, 2.5, 0.5, seed=11, hash_salt=3, ring_capacity=64, games_budget=G, lib_path=emu_lib,
                         eval_cache_log2=cache, **kw)
    recs, faults = [], 0
    for rnd in range(4000):
        if rnd % 3 == 0 and rnd < 60:
            eng.debug_fused_fault(2)                       # games 3..5, 9..11 lose the evaluations of the following waves, until the host notices
        eng.run_waves(5)
        st = eng.stats()                                   # a synchronisation point: faults seen, hook cleared
        faults = st["fused_faults"]
        recs += eng.drain_finished()
        if len(recs) == G:
            break
    eng.close()
    assert len(recs) == G and faults > 0, (len(recs), faults)
    for r in recs:
        if search == "gumbel":
            o = oracle.selfplay_game_gumbel("Connect4", iters, 42, 4, 50.0, 1.0, 11, r["slot"], 0, hash_salt=3)
        else:
            o = oracle.selfplay_game("Connect4", iters, 42, 8, 7, 2.5, 0.5, 11, r["slot"], 0, hash_salt=3)
        assert r["T"] == o["T"], r["slot"]
        for k in ("actions", "root_N", "root_W", "policies", "values"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=f"slot {r['slot']} {k}")


def test_emu_read_positions_is_the_counterpart_of_set_position(emu_lib):
    """gaz_engine_read_positions (round 3): every game's action_history (Guide.py:111-133) as action indices — what gaz_engine_set_position takes.
    A second engine put at those positions continues from exactly them."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine
    a = SelfPlayEngine("Connect4", 6, 16, 42, 8, 7, 2.5, 0.5, seed=3, hash_salt=1, lib_path=emu_lib)
    a.run_waves(150)
    hs = a.read_positions()
    assert len(hs) == 6 and any(len(h) > 2 for h in hs) and all(0 <= x < 7 for h in hs for x in h)
    b = SelfPlayEngine("Connect4", 6, 16, 42, 8, 7, 2.5, 0.5, seed=3, hash_salt=1, lib_path=emu_lib)
    for g, h in enumerate(hs):
        if h:
            b.set_position(g, h)
    b.run_waves(1)
    back = b.read_positions()
    assert all(back[g][:len(hs[g])] == hs[g] for g in range(6))
    a.close(); b.close()


def test_bench_random_histories_are_legal_and_unfinished():
    """bench.py's staggered start for Gomoku: random legal playouts that have not ended."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from grok_alpha_zero_amd.games import GAMES
    for game, span in (("Gomoku", 30), ("Connect4", 20)):
        hs = bench.random_histories(game, 25, np.random.default_rng(5), span)
        assert len(hs) == 25 and max(len(h) for h in hs) <= span
        for h in hs:
            g = GAMES[game]()
            for a in h:
                assert g.check_win() == -2 if g.action_history else True
                g.do_action(GAMES[game].index_to_action(int(a)))
            assert not h or g.check_win() == -2


@pytest.mark.parametrize("search,groups,G,budget", [("puct", 2, 10, 27), ("puct", 3, 11, 11), ("gumbel", 2, 9, 20)])
def test_emu_game_groups_play_exactly_the_games_of_one_batch(emu_lib, oracle, search, groups, G, budget):
    """gaz_engine_config::game_groups (round 3): K groups of consecutive slots, each an engine of its own (stream, batch, launch per wave) behind ONE
    handle.  No game may depend on it: with a games_budget that is no multiple of n_games and uneven group sizes, the grouped engine plays exactly
    the (slot, game_seq) set of one batch, every record equal to the oracle's; counters are sums; per-slot calls reach the right group."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, SEARCH_GUMBEL, SEARCH_PUCT
    iters = 16
    kw = dict(search=SEARCH_GUMBEL, gumbel_m=4, c_visit=50.0, c_scale=1.0) if search == "gumbel" else dict(search=SEARCH_PUCT)
    out = {}
    for k in (1, groups):
        eng = SelfPlayEngine("Connect4", G, iters, 42, 4, 3, 2.5, 0.5, seed=21, hash_salt=5, slot_offset=100, ring_capacity=4 * G, games_budget=budget,
                             game_groups=k, lib_path=emu_lib, **kw)
        recs = []
        for _ in range(4000):
            eng.run_waves(8)
            recs += eng.drain_finished(3)                  # a small max_records: no group's ring may starve
            if len(recs) == budget:
                break
        st = eng.stats()
        assert st["game_groups"] == k and len(recs) == budget
        assert st["game_stats"][2] == budget and st["plies"] == sum(r["T"] for r in recs)
        out[k] = ({(r["slot"], r["game_seq"]): r for r in recs}, st)
        eng.close()
    one, many = out[1][0], out[groups][0]
    assert set(one) == set(many) == {(100 + g, q) for q in range(8) for g in range(G) if q * G + g < budget}
    for key in one:
        for f in ("actions", "root_N", "root_W", "root_P", "policies", "q", "evals", "root_visits", "winner", "T", "values"):
            np.testing.assert_array_equal(np.asarray(one[key][f]), np.asarray(many[key][f]), err_msg=f"{key} {f}")
    assert out[1][1]["evals"] == out[groups][1]["evals"] and out[1][1]["sims"] == out[groups][1]["sims"]
    for (slot, seq) in list(one)[:4]:
        o = (oracle.selfplay_game_gumbel("Connect4", iters, 42, 4, 50.0, 1.0, 21, slot, seq, hash_salt=5) if search == "gumbel"
             else oracle.selfplay_game("Connect4", iters, 42, 4, 3, 2.5, 0.5, 21, slot, seq, hash_salt=5))
        np.testing.assert_array_equal(many[(slot, seq)]["actions"], o["actions"])
        np.testing.assert_array_equal(many[(slot, seq)]["root_N"], o["root_N"])


def test_emu_game_groups_route_per_slot_calls_and_refuse_what_they_cannot_serve(emu_lib):
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_EXTERNAL
    mk = lambda k, **kw: SelfPlayEngine("Connect4", 7, 16, 42, 8, 7, 2.5, 0.5, seed=3, hash_salt=1, game_groups=k, lib_path=emu_lib, **kw)
    a, b = mk(1), mk(3)                                    # groups of 3 | 2 | 2 slots
    hs = [[3], [], [0, 1, 2], [6, 6], [], [2, 3, 4, 5], [1]]
    for e in (a, b):
        for g, h in enumerate(hs):
            if h:
                e.set_position(g, h)
        e.run_waves(40)
    pa, pb = a.read_positions(), b.read_positions()
    assert pa == pb and all(pb[g][:len(hs[g])] == hs[g] for g in range(7))
    sa, sb = a.root_stats(), b.root_stats()
    for k in sa:
        np.testing.assert_array_equal(np.asarray(sa[k]), np.asarray(sb[k]), err_msg=k)
    b.reset_games([1, 4, 6])
    a.reset_games([1, 4, 6])
    a.run_waves(3); b.run_waves(3)
    assert a.read_positions() == b.read_positions()
    with pytest.raises(RuntimeError, match="game_groups"):
        b.wave_begin()
    a.close(); b.close()
    with pytest.raises(RuntimeError, match="game_groups"):
        mk(2, sync_moves=True)
    with pytest.raises(RuntimeError, match="game_groups"):
        mk(2, evaluator=EVAL_EXTERNAL)
    with pytest.raises(RuntimeError, match="game_groups"):
        mk(8)
    with pytest.raises(RuntimeError, match="games_budget"):
        mk(2, games_budget=5)
