"""Soak check of the fused tree + trunk launch (k_wave_trunk) at the headline size: the same long run twice — once fused (leaf rows handed
from tree teams to trunk workgroups through system-scope stores / loads inside a running kernel), once as separate launches — must
finish exactly the same games with exactly the same records.  A stale plane or a missed flag would show up as a diverging game.
    python tools/soak_fused.py [games=4096] [waves=6000] [cache_log2=0]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET
from grok_alpha_zero_amd.net import Connect4Net

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
cache = int(sys.argv[3]) if len(sys.argv) > 3 else 0
config = sys.argv[4] if len(sys.argv) > 4 else "connect4"       # connect4 | gumbel (8192 games, n = 32, m = 7) | gomoku (2048 games, 400 sims, games cut at 8 plies)
from grok_alpha_zero_amd.engine import SEARCH_GUMBEL
from grok_alpha_zero_amd.net import NETS
if config == "gumbel":
    w = Connect4Net(6, seed=0, policy_head="linear").eval().export_engine_weights()
    mk = lambda groups: SelfPlayEngine("Connect4", G, 32, 42, 8, 7, 2.5, 0.5, seed=1234, evaluator=EVAL_RESNET, net_blocks=6, ring_capacity=4 * G, search=SEARCH_GUMBEL,
                                       gumbel_m=7, c_visit=50.0, c_scale=1.0, policy_is_logits=True, game_groups=groups)
elif config == "gomoku":
    w = NETS["Gomoku"](10, seed=0).eval().export_engine_weights()
    mk = lambda groups: SelfPlayEngine("Gomoku", G, 400, 8, 6, 4, 4.5, 0.05, seed=77, evaluator=EVAL_RESNET, net_blocks=10, net_filters=128, ring_capacity=4 * G, game_groups=groups)
else:
    w = Connect4Net(6, seed=0).eval().export_engine_weights()
    mk = lambda groups: SelfPlayEngine("Connect4", G, 200, 42, 8, 7, 2.5, 0.5, seed=1234, evaluator=EVAL_RESNET, net_blocks=6, ring_capacity=4 * G, eval_cache_log2=cache,
                                       game_groups=groups)
out = []
# (round 3) three runs: the default (game groups where the library chooses them: two fused launches in flight together), one batch fused, and the
# default's grouping with separate launches
for fused, groups in ((True, 0), (True, 1), (False, 0)):
    eng = mk(groups)
    eng.load_weights(w); eng.set_fused_wave(fused)
    t0 = time.time(); recs = {}
    for _ in range(waves // 200):
        eng.run_waves(200)
        for r in eng.drain_finished():
            h = hashlib.sha256()
            for k in ("actions", "root_N", "root_W", "root_P", "policies", "q", "evals", "root_visits"):
                h.update(np.ascontiguousarray(r[k]).tobytes())
            recs[(r["slot"], r["game_seq"])] = (r["T"], r["winner"], h.hexdigest())
    st = eng.stats(); eng.close()
    print(f"fused={fused} groups={st['game_groups']}: {len(recs)} games finished in {time.time() - t0:.1f} s, {st['evals']} evaluations ({st['cache_hits']} cache hits), "
          f"fused flag {st['fused_wave']}, workgroups that gave up {st['fused_faults']}", flush=True)
    out.append(recs)
a = out[0]
fail = False
for name, b in (("one batch, fused", out[1]), ("separate launches", out[2])):
    common = set(a) & set(b)
    bad = [k for k in common if a[k] != b[k]]
    print(f"default vs {name}: {len(common)} games in both runs, {len(set(a) ^ set(b))} only in one (finished in the last launches), mismatching records: {len(bad)}")
    fail = fail or bool(bad) or len(common) < 100
sys.exit(1 if fail else 0)
