"""Soak check: many finished games of a long free-running engine (evaluation cache on) against the CPU oracle, game by game."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grok_alpha_zero_amd.engine import SelfPlayEngine, SEARCH_GUMBEL, SEARCH_PUCT
from oracle import gaz_oracle as O

def run(game, search, G, iters, max_actions, waves, cache, check):
    gum = search == SEARCH_GUMBEL
    eng = SelfPlayEngine(game, G, iters, max_actions, 4, 3, 2.5, 0.5, seed=4242, hash_salt=9, ring_capacity=8 * G, search=search,
                         gumbel_m=7 if gum else 0, c_visit=50.0, c_scale=1.0, eval_cache_log2=cache)
    recs = []
    for _ in range(waves // 200):
        eng.run_waves(200); recs += eng.drain_finished()
    st = eng.stats(); eng.close()
    rng = np.random.default_rng(1); pick = rng.permutation(len(recs))[:check]
    bad = 0
    for i in pick:
        r = recs[i]
        if gum:
            o = O.selfplay_game_gumbel(game, iters, max_actions, 7, 50.0, 1.0, 4242, r["slot"], r["game_seq"], hash_salt=9)
        else:
            o = O.selfplay_game(game, iters, max_actions, 4, 3, 2.5, 0.5, 4242, r["slot"], r["game_seq"], hash_salt=9)
        ok = r["T"] == o["T"] and all(np.array_equal(r[k], o[k][:r["T"]] if o[k].shape[0] != r[k].shape[0] else o[k]) for k in ("actions", "root_N", "root_W", "policies"))
        bad += not ok
    print(f"{game} {'gumbel' if gum else 'puct'} cache=2^{cache}: {len(recs)} games finished, {st['cache_hits']} hits / {st['evals']} requests, checked {len(pick)} vs oracle, mismatches {bad}", flush=True)
    return bad

t0 = time.time(); bad = 0
# round 2: without the evaluation cache the PUCT kernel of the small boards runs four games per wavefront (16-lane teams)
bad += run("Connect4", SEARCH_PUCT, 1024, 60, 42, 6000, 0, 600)
bad += run("TicTacToe", SEARCH_PUCT, 512, 30, 9, 2000, 0, 600)
bad += run("Connect4", SEARCH_PUCT, 1024, 60, 42, 6000, 20, 600)
bad += run("Connect4", SEARCH_GUMBEL, 1024, 32, 42, 4000, 20, 600)
bad += run("TicTacToe", SEARCH_PUCT, 512, 30, 9, 2000, 16, 600)
bad += run("Gomoku", SEARCH_PUCT, 64, 700, 6, 6000, 16, 60)      # run_iterations < 225 legal moves would become 3 x legal anyway (MCTS.py:545)
print("soak done in %.0f s, total mismatches %d" % (time.time() - t0, bad))
sys.exit(1 if bad else 0)
