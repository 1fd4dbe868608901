"""Search + Self_Play with a REAL network behind session.run, against fixtures recorded from the reference (tools/gen_golden.py
NET_CASES: the reference's Self_Play.play() driven with a random-init ResNet — net.py's fp32 restatement — as the session).  The
fixture keeps every (input state -> policy, value) pair the session answered, so all three implementations consume the SAME
evaluator outputs through their own session boundary:
  oracle   evaluator callback                                         (CPU suite)
  engine   GAZ_EVAL_EXTERNAL: wave_begin / read_batch / write_outputs  (emulation build in the CPU suite, HIP under -m gpu)
Softmax-probability priors (PUCT) and raw-logit priors (Gumbel) flow through make_priors / the Gumbel math exactly as in the
reference; N, W, P, policies, values must match bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CASES = ["c4_netp_a", "ttt_netp_a", "c4_netg_a"]
EMU_DIR = os.path.join(ROOT, "tests", "emu")


def _table(fx):
    return {s.tobytes(): (p, v) for s, p, v in zip(fx["eval_states"], fx["eval_policy"], fx["eval_value"])}


def _compare(r, fx):
    assert r["T"] == len(fx["actions"])
    for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies"):
        np.testing.assert_array_equal(r[k], fx[k], err_msg=k)
    np.testing.assert_array_equal(np.asarray(r["values"]).reshape(-1), fx["values"].reshape(-1))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_with_a_real_network(oracle, name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    tab = _table(fx)
    calls = [0]

    def ev(state):
        calls[0] += 1
        return tab[np.ascontiguousarray(state, np.int8).tobytes()]        # KeyError = the oracle asked for a state the reference never evaluated
    if int(fx["is_gumbel"]):
        o = oracle.selfplay_game_gumbel(str(fx["game"]), int(fx["run_iterations"]), int(fx["max_actions"]), int(fx["m"]), float(fx["c_visit"]),
                                        float(fx["c_scale"]), int(fx["seed"]), int(fx["slot"]), int(fx["game_seq"]), evaluator=ev)
    else:
        o = oracle.selfplay_game(str(fx["game"]), int(fx["run_iterations"]), int(fx["max_actions"]), int(fx["explore_first"]),
                                 int(fx["explore_second"]), float(fx["c_puct_init"]), float(fx["dirichlet_alpha"]), int(fx["seed"]),
                                 int(fx["slot"]), int(fx["game_seq"]), evaluator=ev)
    _compare(o, fx)
    np.testing.assert_array_equal(o["states"], fx["states"])
    assert calls[0] == int(fx["evaluator_calls"])


def _engine_external(fx, lib_path):
    """the INTEGRATION.md section 3 loop: one batched evaluator call per wave at the session boundary"""
    from grok_alpha_zero_amd.engine import EVAL_EXTERNAL, SEARCH_GUMBEL, SEARCH_PUCT, SelfPlayEngine
    gum = bool(int(fx["is_gumbel"]))
    G = 3                                                                # slot 0 of this engine is the fixture's slot; the others just play along
    eng = SelfPlayEngine(str(fx["game"]), G, int(fx["run_iterations"]), int(fx["max_actions"]), int(fx["explore_first"]), int(fx["explore_second"]),
                         float(fx["c_puct_init"]), float(fx["dirichlet_alpha"]), int(fx["seed"]), slot_offset=int(fx["slot"]),
                         evaluator=EVAL_EXTERNAL, ring_capacity=16, search=SEARCH_GUMBEL if gum else SEARCH_PUCT,
                         gumbel_m=int(fx["m"]) if gum else 0, c_visit=float(fx["c_visit"]) if gum else 50.0, c_scale=float(fx["c_scale"]) if gum else 1.0,
                         first_game_seq=int(fx["game_seq"]), games_budget=G,     # every slot plays exactly one game, then halts
                         lib_path=lib_path)
    tab = _table(fx)
    A = fx["eval_policy"].shape[1]
    uniform = np.full(A, 1.0 / A, np.float32)
    want = int(fx["slot"])
    n_lookups = 0
    rec = None
    for _ in range(200000):
        eng.wave_begin()
        x, pend = eng.read_batch()
        pol = np.zeros((G, A), np.float32); val = np.zeros(G, np.float32)
        for g in range(G):
            if not pend[g]:
                continue
            if g == 0 and rec is None:
                pol[g], val[g] = tab[x[g].tobytes()]; n_lookups += 1
            else:
                pol[g] = uniform                                          # bystanders (and slot 0's later games) get a dummy evaluator
        eng.write_outputs(pol, val)
        for r in eng.drain_finished():
            if r["slot"] == want and r["game_seq"] == int(fx["game_seq"]) and rec is None:
                rec = r
        if rec is not None:
            break
    eng.close()
    assert rec is not None, "the game did not finish"
    assert n_lookups == int(fx["evaluator_calls"])
    return rec


@pytest.mark.parametrize("name", CASES)
def test_external_evaluator_engine_matches_reference_emu(name):
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    _compare(_engine_external(fx, os.path.join(EMU_DIR, "libgaz_emu.so")), fx)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_external_evaluator_engine_matches_reference_hip(name):
    """GAZ_EVAL_EXTERNAL on the MI355X (this mode had only ever run on the emulation build): the tree kernels on the GPU,
    the evaluator outputs written by the host between waves, results equal to the reference's Self_Play.play() with that network."""
    import torch
    assert torch.cuda.is_available()
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    _compare(_engine_external(fx, None), fx)


@pytest.mark.parametrize("name", CASES)
def test_fixture_evaluator_is_the_restated_network(name):
    """The (state -> policy, value) table in the fixture really is net.py's network of the stated seed (fp32, within float noise of
    a different thread count / machine), i.e. the reference was driven by the ResNet, not by some other function."""
    import torch
    from grok_alpha_zero_amd.net import NETS
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    net = NETS[str(fx["game"])](int(fx["net_blocks"]), seed=int(fx["net_seed"]), policy_head="linear" if int(fx["is_gumbel"]) else "softmax").eval()
    idx = np.linspace(0, len(fx["eval_value"]) - 1, 40).astype(int)
    with torch.no_grad():
        p, v = net(torch.from_numpy(fx["eval_states"][idx].copy()))
    np.testing.assert_allclose(p.numpy(), fx["eval_policy"][idx], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(v.numpy().reshape(-1), fx["eval_value"][idx], rtol=1e-4, atol=1e-6)
    if not int(fx["is_gumbel"]):
        assert np.allclose(fx["eval_policy"].sum(1), 1.0, atol=1e-5)      # probabilities for PUCT, raw logits for Gumbel (Build_Model.py:54-60)
