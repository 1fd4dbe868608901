"""-m gpu: every BASELINE configuration at its REAL composition against the oracle.

The other GPU tests check the pieces — the full-size search with the hash evaluator, the network per layer, the fused launch against
separate launches.  Here the thing that is benchmarked is the thing that is checked: the full batch, the full network, the default
launch shape (Connect4: tree step + trunk as ONE launch with the edge tiles; Gomoku: the 10-block launch with block 0 inside; Gumbel:
logits head), the configuration's own search parameters — and sampled slots are replayed by the CPU oracle (oracle/gaz_puct.c,
gaz_gumbel.c: MCTS.py / MCTS_Gumbel.py / Self_Play.py restated) whose session.run is served by the SAME HIP network on single rows
(rows of a batch are independent bit for bit, tests/test_evaluator_gpu.py).  Bar: actions, visit counts N, W, P, policies and values
bit-exact (north_star: integer visit counts bit-exact, Q / V within 1e-5 — here equal)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _first_games(eng, n_games, max_rounds, waves):
    recs = []
    for _ in range(max_rounds):
        eng.run_waves(waves)
        recs += eng.drain_finished(n_games)
        if len(recs) >= n_games:
            break
    return {r["slot"]: r for r in recs if r["game_seq"] == 0}


def _check(r, o, what):
    assert r["T"] == o["T"], what
    for k in ("actions", "root_N", "root_visits", "root_W", "root_P", "policies", "values"):
        np.testing.assert_array_equal(np.asarray(r[k]), np.asarray(o[k]), err_msg=f"{what}: {k}")


def test_headline_composition_connect4_4096_games_6_blocks_fused_launch(oracle):
    """BASELINE configs[1] exactly as bench.py runs it: 4096 concurrent Connect4 games, 200 simulations per move, the 6-block network in
    bf16, tree step + trunk in ONE launch (k_wave_trunk<mix, edge tiles>), tau schedule 8 / 7, Dirichlet 0.5 — every slot plays its first
    game to the end (Self_Play.py:71-157); 10 sampled slots (first / last board of a 3-board tile, the 96-row tiles of the last round, both
    ends of the batch) are replayed by the oracle with the HIP network as its evaluator."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
    from grok_alpha_zero_amd.net import Connect4Net
    G, sims = 4096, 200
    net = Connect4Net(6, seed=0).eval()
    w = net.export_engine_weights()
    eng = SelfPlayEngine("Connect4", G, sims, 42, 8, 7, 2.5, 0.5, seed=1234, evaluator=EVAL_RESNET, net_blocks=6, ring_capacity=G, games_budget=G)
    eng.load_weights(w)
    first = _first_games(eng, G, 200, 400)
    st = eng.stats()
    assert len(first) == G and st["fused_wave"] == 1 and st["fused_faults"] == 0, (len(first), st)
    eng.close()
    probe = SelfPlayEngine("Connect4", 64, 1, 42, 8, 7, 2.5, 0.5, seed=0, evaluator=EVAL_RESNET, net_blocks=6, ring_capacity=0)
    probe.load_weights(w)

    def ev(state):
        p, v, _ = probe.evaluate(state[None])
        return p[0], v[0]
    for slot in (0, 2, 3, 1535, 1536, 3071, 3072, 3073, 4000, 4095):
        o = oracle.selfplay_game("Connect4", sims, 42, 8, 7, 2.5, 0.5, 1234, slot, 0, evaluator=ev)
        _check(first[slot], o, f"Connect4 headline composition, slot {slot}")
    probe.close()
    lengths = np.array([r["T"] for r in first.values()])
    assert 7 <= lengths.min() and lengths.max() <= 42


def test_gomoku_composition_2048_games_10_blocks(oracle):
    """BASELINE configs[3]: 2048 concurrent Gomoku games, 400 simulations per move (on an empty 15 x 15 board MCTS.run raises that to 3 x the
    legal moves, MCTS.py:545-546), the 10-block network (256-channel stem, block 0 with its projection inside the 8-wave trunk launch, heads on
    two streams), alpha 0.05, c_puct 4.5 — max_actions = 3 keeps the oracle's side to seconds (a game is then a draw after 3 plies,
    Self_Play.py:155-157); 4 sampled slots replayed by the oracle with the HIP network as its evaluator."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
    from grok_alpha_zero_amd.net import NETS
    G, sims, plies = 2048, 400, 3
    net = NETS["Gomoku"](10, seed=0).eval()
    w = net.export_engine_weights()
    eng = SelfPlayEngine("Gomoku", G, sims, plies, 6, 4, 4.5, 0.05, seed=77, evaluator=EVAL_RESNET, net_blocks=10, net_filters=128, ring_capacity=G, games_budget=G)
    eng.load_weights(w)
    first = _first_games(eng, G, 60, 100)
    assert len(first) == G, len(first)
    eng.close()
    probe = SelfPlayEngine("Gomoku", 8, 1, plies, 6, 4, 4.5, 0.05, seed=0, evaluator=EVAL_RESNET, net_blocks=10, net_filters=128, ring_capacity=0)
    probe.load_weights(w)

    def ev(state):
        p, v, _ = probe.evaluate(state[None])
        return p[0], v[0]
    for slot in (0, 1, 1023, 2047):
        o = oracle.selfplay_game("Gomoku", sims, plies, 6, 4, 4.5, 0.05, 77, slot, 0, evaluator=ev)
        _check(first[slot], o, f"Gomoku composition, slot {slot}")
    probe.close()


def test_gumbel_composition_8192_games_logits_head(oracle):
    """BASELINE configs[4]: 8192 concurrent Connect4 games under the Gumbel search (n = 32, m = 7, c_visit 50, c_scale 1, Gumbel noise on;
    MCTS_Gumbel.py:562-679) with the 6-block network's LOGITS head behind it (Connect4/Build_Model.py:54-60) — whole games; 8 sampled slots
    replayed by the oracle with the HIP network as its evaluator."""
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET, SEARCH_GUMBEL
    from grok_alpha_zero_amd.net import Connect4Net
    G, n, m = 8192, 32, 7
    net = Connect4Net(6, seed=0, policy_head="linear").eval()
    w = net.export_engine_weights()
    eng = SelfPlayEngine("Connect4", G, n, 42, 8, 7, 2.5, 0.5, seed=4321, evaluator=EVAL_RESNET, net_blocks=6, ring_capacity=G, games_budget=G,
                         search=SEARCH_GUMBEL, gumbel_m=m, c_visit=50.0, c_scale=1.0, policy_is_logits=True)
    eng.load_weights(w)
    first = _first_games(eng, G, 100, 200)
    assert len(first) == G, len(first)
    eng.close()
    probe = SelfPlayEngine("Connect4", 64, 1, 42, 8, 7, 2.5, 0.5, seed=0, evaluator=EVAL_RESNET, net_blocks=6, ring_capacity=0, policy_is_logits=True)
    probe.load_weights(w)

    def ev(state):
        p, v, _ = probe.evaluate(state[None])
        return p[0], v[0]
    for slot in (0, 1, 1536, 4095, 4096, 6000, 8190, 8191):
        o = oracle.selfplay_game_gumbel("Connect4", n, 42, m, 50.0, 1.0, 4321, slot, 0, evaluator=ev)
        _check(first[slot], o, f"Gumbel composition, slot {slot}")
    probe.close()
