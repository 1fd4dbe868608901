"""Generate tests/golden/*.npz from the REFERENCE itself (imported under tools/ref_shim.py, in the build
container only).  Each fixture holds inputs (config, seed) and the reference's outputs: per ply the root
statistics MCTS.run returned (N, W, P by policy index, root.visits), the move played, and the final
Self_Play arrays (input states, improved policies, q, z, values, game_stats) — data only.

    python tools/gen_golden.py            # regenerate everything
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tools import ref_shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

GAME_CLASS = {"TicTacToe": ("TicTacToe", "TicTacToe"), "Connect4": ("Connect4", "Connect4"), "Gomoku": ("Gomoku", "Gomoku")}


def action_to_index(game, action):
    if game == "Connect4":
        return int(action)
    W = 3 if game == "TicTacToe" else 15
    x, y = int(action[0]), int(action[1])
    return y * W + x


def ref_selfplay_puct(game, iteration_limit, max_actions, explore_first, explore_second, c_puct_init, alpha,
                      seed, slot, game_seq, salt, session=None, gumbel=None, opening=None):
    """Run the reference's Self_Play.play() once; returns the fixture dict."""
    ref = ref_shim.load_reference()
    inj = ref_shim.activate(seed, slot, game_seq)
    cls = getattr(ref[GAME_CLASS[game][0]], GAME_CLASS[game][1])
    g = cls()
    A = g.policy_shape[0]
    sess = session if session is not None else ref_shim.HashSession(A, salt)
    train_config = {"MCTS_iteration_limit": iteration_limit, "MCTS_time_limit": None, "use_gumbel": False,
                    "use_njit": False, "c_puct_init": c_puct_init, "dirichlet_alpha": alpha,
                    "max_actions": max_actions, "num_explore_actions_first": explore_first,
                    "num_explore_actions_second": explore_second}
    build_config = {}
    if gumbel is not None:
        train_config.update(use_gumbel=True, m=gumbel[0], c_visit=gumbel[1], c_scale=gumbel[2])
        if len(gumbel) > 3 and gumbel[3]:
            build_config["use_stablemax"] = True           # MCTS_Gumbel(activation_fn="stablemax") (Self_Play.py:69)
    if opening is not None:
        train_config["opening_actions"] = opening
    folder = f"/fake/{game}_{iteration_limit}_{seed}_{slot}_{game_seq}/1"
    ref_shim._FakeH5File.STORE.pop(folder + "/Self_Play_Data.h5", None)
    f = ref_shim._FakeH5File(folder + "/Self_Play_Data.h5")
    f.create_dataset("game_stats", data=np.zeros(6, np.uint32), dtype=np.uint32)

    class _Lock:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    sp = ref["Self_Play"].Self_Play(g, sess, build_config, train_config, _Lock(), folder, 1)
    sp.play()
    T = len(g.action_history)
    rN = np.zeros((T, A), np.uint32); rW = np.zeros((T, A), np.float32); rP = np.zeros((T, A), np.float32)
    rV = np.zeros(T, np.uint64); acts = np.zeros(T, np.int32)
    assert len(inj.run_log) == T
    for t, (tree, (move, rows)) in enumerate(inj.run_log):
        acts[t] = action_to_index(game, move)
        for r in rows:
            a = action_to_index(game, r[0])
            rN[t, a] = r[4]; rW[t, a] = r[3]; rP[t, a] = r[5]
        rV[t] = int(rows[0][6])
    search_acts = acts.copy()
    acts = np.array([action_to_index(game, a) for a in g.action_history], np.int32)    # what was PLAYED (differs at move 0 with opening_actions)
    d = f.d
    n_aug = (len(d) - 1) // 3
    out = dict(game=game, iteration_limit=iteration_limit, run_iterations=int(iteration_limit * 1.5),
               max_actions=max_actions, explore_first=explore_first, explore_second=explore_second,
               c_puct_init=c_puct_init, dirichlet_alpha=alpha, seed=seed, slot=slot, game_seq=game_seq, salt=salt,
               actions=acts, search_actions=search_acts, root_N=rN, root_W=rW, root_P=rP, root_visits=rV,
               states=d["boards_0"].data, policies=d["policies_0"].data, values=d["values_0"].data,
               game_stats=d["game_stats"].data, n_aug=n_aug, evaluator_calls=sess.calls)
    if opening is not None:
        out.update(opening_idx=np.array([action_to_index(game, a) for a, _ in opening], np.int32), opening_w=np.array([w for _, w in opening]))
    if gumbel is not None:
        out.update(m=gumbel[0], c_visit=gumbel[1], c_scale=gumbel[2], run_iterations=iteration_limit,
                   stablemax=int(len(gumbel) > 3 and bool(gumbel[3])))
    for k in range(n_aug):
        out[f"aug_boards_{k}"] = d[f"boards_{k}"].data
        out[f"aug_policies_{k}"] = d[f"policies_{k}"].data
    return out


class NetSession:
    """session.run backed by a REAL network: the fp32 PyTorch restatement of the reference's ResNet (grok_alpha_zero_amd/net.py,
    random-init, one thread so the bits do not depend on the machine's core count), batch of one, exactly where the reference's
    MCTS calls it (MCTS.py:224-235).  Every (input state -> policy, value) pair it answered is kept, so the oracle and the engine
    can be fed the SAME evaluator outputs through their own session boundary (oracle callback / GAZ_EVAL_EXTERNAL)."""

    def __init__(self, game, blocks, policy_head, net_seed):
        import torch
        from grok_alpha_zero_amd.net import NETS
        torch.set_num_threads(1)
        self.torch = torch
        self.net = NETS[game](blocks, seed=net_seed, policy_head=policy_head).eval()
        self.calls, self.table = 0, {}

    def run(self, output_names, input_feed, **kw):
        x = np.ascontiguousarray(input_feed["inputs"][0]).astype(np.int8)
        self.calls += 1
        key = x.tobytes()
        if key not in self.table:
            with self.torch.no_grad():
                p, v = self.net(self.torch.from_numpy(x[None].copy()))
            self.table[key] = (p[0].numpy().astype(np.float32).copy(), np.float32(v.reshape(-1)[0].item()))
        p, v = self.table[key]
        return [p.reshape(1, -1).copy(), np.array([[v]], np.float32)]

    def arrays(self, shape):
        keys = list(self.table)
        return dict(eval_states=np.array([np.frombuffer(k, np.int8).reshape(shape) for k in keys]),
                    eval_policy=np.array([self.table[k][0] for k in keys]), eval_value=np.array([self.table[k][1] for k in keys], np.float32))


NET_CASES = [   # Self_Play.play() with a real random-init ResNet behind session.run ("hash evaluator AND a small random-init ResNet")
    # name, game, limit, max_actions, ef, es, c_puct, alpha, seed, slot, seq, blocks, net_seed, gumbel
    ("c4_netp_a", "Connect4", 40, 42, 8, 7, 2.5, 0.5, 501, 0, 0, 2, 3, None),
    ("ttt_netp_a", "TicTacToe", 34, 9, 2, 1, 1.25, 1.0, 502, 1, 0, 2, 4, None),
    ("c4_netg_a", "Connect4", 32, 42, 0, 0, 0.0, 0.0, 503, 2, 0, 2, 5, (7, 50.0, 1.0)),
]


PUCT_CASES = [
    # name, game, MCTS_iteration_limit, max_actions, explore_first, explore_second, c_puct_init, alpha, seed, slot, seq, salt
    ("ttt_puct_a", "TicTacToe", 34, 9, 2, 1, 1.25, 1.0, 1234, 0, 0, 7),       # int(34*1.5) = 51 ~ config[0]'s 50 sims
    ("ttt_puct_b", "TicTacToe", 34, 9, 2, 1, 1.25, 1.0, 1234, 3, 1, 7),
    ("c4_puct_a", "Connect4", 40, 42, 8, 7, 2.5, 0.5, 1234, 0, 0, 11),
    ("c4_puct_b", "Connect4", 40, 42, 8, 7, 2.5, 0.5, 1234, 5, 2, 11),
    ("c4_puct_c", "Connect4", 134, 42, 8, 7, 2.5, 0.5, 99, 17, 0, 3),           # int(134*1.5) = 201 sims (headline n=200)
    ("gmk_puct_a", "Gomoku", 40, 12, 6, 4, 4.5, 0.05, 1234, 0, 0, 5),
    # Gomoku games that run to a NATURAL end (five in a row inside a real game; max_actions = the whole board).  int(150 * 1.5) = 225
    # simulations per move = the fewest MCTS.run accepts on an empty 15 x 15 board (iteration_limit < n_legal -> 3 n_legal, MCTS.py:545-546)
    ("gmk_puct_win_a", "Gomoku", 150, 225, 6, 4, 4.5, 0.05, 78, 1, 0, 9),
    ("gmk_puct_win_b", "Gomoku", 150, 225, 6, 4, 4.5, 0.05, 4242, 6, 1, 2),
]


def ref_single_tree_puct(game, iteration_limit, c_puct_init, alpha, seed, salt, max_plies, fast_find_win=False, taus=None, updates=None):
    """The reference's MCTS class used on its own (Connect4/play.py, Game_Tester.py:480-513): ONE tree searches every move.
    taus: tau per ply (general tau, MCTS.py:606-610); updates: {ply: kwargs} passed to update_hyperparams before that ply's run."""
    ref = ref_shim.load_reference()
    inj = ref_shim.activate(seed, 0, 0)
    cls = getattr(ref[GAME_CLASS[game][0]], GAME_CLASS[game][1])
    g = cls(); A = g.policy_shape[0]
    sess = ref_shim.HashSession(A, salt)
    mcts = ref["MCTS"].MCTS(g, sess, use_njit=False, c_puct_init=c_puct_init, use_dirichlet=True, dirichlet_alpha=alpha,
                            dirichlet_epsilon=0.25, tau=1.0, fast_find_win=fast_find_win)
    acts, rN, rW, rP, rV, rT = [], [], [], [], [], []
    for ply in range(max_plies):
        mcts.update_hyperparams(tau=(1.0 if ply < 4 else 0) if taus is None else taus[ply % len(taus)])
        if updates and ply in updates:
            mcts.update_hyperparams(**updates[ply])
        move, rows = mcts.run(iteration_limit=iteration_limit, use_bar=False)
        N = np.zeros(A, np.uint32); Wv = np.zeros(A, np.float32); P = np.zeros(A, np.float32); Tm = np.full(A, -9, np.int32)
        for r in rows:
            a = action_to_index(game, r[0]); N[a] = r[4]; Wv[a] = r[3]; P[a] = r[5]; Tm[a] = -2 if r[7] is None else int(r[7])
        acts.append(action_to_index(game, move)); rN.append(N); rW.append(Wv); rP.append(P); rV.append(int(rows[0][6])); rT.append(Tm)
        g.do_action(move)
        if g.check_win() != -2:
            break
        mcts.prune_tree(move)
    return dict(game=game, iteration_limit=iteration_limit, c_puct_init=c_puct_init, dirichlet_alpha=alpha, seed=seed, salt=salt,
                actions=np.array(acts, np.int32), root_N=np.array(rN), root_W=np.array(rW), root_P=np.array(rP),
                root_visits=np.array(rV, np.uint64), evaluator_calls=sess.calls, fast_find_win=int(fast_find_win),
                is_terminal=np.array(rT), taus=np.array(taus if taus is not None else [], np.float64),
                update_plies=np.array(sorted(updates) if updates else [], np.int32),
                update_json=np.array(__import__("json").dumps({str(k): v for k, v in (updates or {}).items()})))


def ref_single_gumbel(game, iteration_limit, m, c_visit, c_scale, seed, salt, max_plies, use_gumbel_noise, updates=None):
    """The reference's MCTS_Gumbel class on its own with use_gumbel_noise as given (the class default is False, MCTS_Gumbel.py:157);
    a fresh object per move like Self_Play.py:151-153 — one continuing injected stream."""
    ref = ref_shim.load_reference()
    ref_shim.activate(seed, 0, 0)
    cls = getattr(ref[GAME_CLASS[game][0]], GAME_CLASS[game][1])
    g = cls(); A = g.policy_shape[0]
    sess = ref_shim.HashSession(A, salt)
    acts, pis, rN, rW, rP, rV = [], [], [], [], [], []
    kw = dict(m=m, c_visit=c_visit, c_scale=c_scale)
    for ply in range(max_plies):
        mcts = ref["MCTS_Gumbel"].MCTS_Gumbel(g, sess, use_gumbel_noise=use_gumbel_noise, use_njit=False, **kw)
        if updates and ply in updates:
            mcts.update_hyperparams(**updates[ply]); kw.update(updates[ply])
        move, rows = mcts.run(iteration_limit=iteration_limit, use_bar=False)
        N = np.zeros(A, np.uint32); Wv = np.zeros(A, np.float32); P = np.zeros(A, np.float32); pi = np.zeros(A, np.float32)
        for r in rows:
            a = action_to_index(game, r[0]); N[a] = r[4]; Wv[a] = r[3]; P[a] = r[5]; pi[a] = r[1]
        acts.append(action_to_index(game, move)); rN.append(N); rW.append(Wv); rP.append(P); pis.append(pi); rV.append(int(rows[0][6]))
        g.do_action(move)
        if g.check_win() != -2:
            break
    return dict(game=game, iteration_limit=iteration_limit, m=m, c_visit=c_visit, c_scale=c_scale, seed=seed, salt=salt,
                use_gumbel_noise=int(use_gumbel_noise), actions=np.array(acts, np.int32), policies=np.array(pis), root_N=np.array(rN),
                root_W=np.array(rW), root_P=np.array(rP), root_visits=np.array(rV, np.uint64), evaluator_calls=sess.calls,
                update_plies=np.array(sorted(updates) if updates else [], np.int32),
                update_json=np.array(__import__("json").dumps({str(k): v for k, v in (updates or {}).items()})))


SINGLE_GUMBEL_CASES = [   # name, game, n, m, c_visit, c_scale, seed, salt, max_plies, use_gumbel_noise, updates
    ("c4_gsingle_nonoise", "Connect4", 32, 7, 50.0, 1.0, 31, 4, 42, False, None),
    ("ttt_gsingle_nonoise", "TicTacToe", 16, 4, 50.0, 2.0, 32, 5, 9, False, None),
    ("c4_gsingle_update", "Connect4", 40, 7, 50.0, 1.0, 33, 6, 10, True, {3: dict(m=4, c_scale=0.5), 6: dict(c_visit=20.0)}),
]


TAU_CASES = [   # general tau N^(1/tau) (MCTS.py:606-610) and update_hyperparams of the PUCT constants between moves (MCTS.py:134-168)
    ("c4_mcts_single_tau", "Connect4", 80, 2.5, 0.5, 41, 12, 42, False, [0.5, 2.0, 0.25, 1.0, 0.8, 3.0, 0.1, 1.5], None),
    ("ttt_mcts_single_tau", "TicTacToe", 40, 1.25, 1.0, 42, 5, 9, False, [0.7, 0.3, 2.5], None),
    ("c4_mcts_single_update", "Connect4", 70, 2.5, 0.5, 43, 8, 14, False, None,
     {2: dict(c_puct_init=1.0), 4: dict(dirichlet_alpha=0.3, dirichlet_epsilon=0.1), 6: dict(c_puct_base=500.0)}),
]

SINGLE_CASES = [("c4_mcts_single", "Connect4", 60, 2.5, 0.5, 21, 9, 42), ("ttt_mcts_single", "TicTacToe", 30, 1.25, 1.0, 22, 4, 9),
                # fast_find_win=True (MCTS.py:88,282-283): a node with a winning move keeps only the first one
                ("c4_mcts_single_ffw", "Connect4", 90, 2.5, 0.5, 23, 6, 42, True), ("ttt_mcts_single_ffw", "TicTacToe", 40, 1.25, 1.0, 24, 3, 9, True)]

GUMBEL_CASES = [
    # name, game, MCTS_iteration_limit, max_actions, m, c_visit, c_scale, seed, slot, seq, salt
    ("ttt_gumbel_a", "TicTacToe", 16, 9, 4, 50.0, 2.0, 1234, 0, 0, 7),
    ("c4_gumbel_a", "Connect4", 32, 42, 7, 50.0, 1.0, 1234, 0, 0, 11),        # BASELINE config 5: n = 32, m = 7
    ("c4_gumbel_b", "Connect4", 32, 42, 7, 50.0, 1.0, 77, 9, 3, 5),
    ("c4_gumbel_c", "Connect4", 64, 42, 4, 50.0, 1.0, 5, 2, 0, 3),
    ("gmk_gumbel_a", "Gomoku", 48, 10, 16, 50.0, 1.0, 1234, 0, 0, 5),
    # activation_fn = "stablemax" (build_config["use_stablemax"]); the trailing True selects it
    ("c4_gumbel_stable_a", "Connect4", 32, 42, 7, 50.0, 1.0, 4321, 1, 0, 13, True),
    ("c4_gumbel_stable_b", "Connect4", 48, 42, 4, 50.0, 0.5, 8, 6, 2, 2, True),
    ("ttt_gumbel_stable_a", "TicTacToe", 16, 9, 4, 50.0, 2.0, 99, 2, 1, 7, True),
    ("gmk_gumbel_stable_a", "Gomoku", 40, 8, 16, 50.0, 1.0, 17, 3, 0, 9, True),
    # Gomoku Gumbel games to a natural end (max_actions = the whole board)
    ("gmk_gumbel_win_a", "Gomoku", 16, 225, 4, 50.0, 1.0, 77, 0, 0, 9),
    ("gmk_gumbel_win_b", "Gomoku", 48, 225, 16, 50.0, 1.0, 91, 5, 0, 3),
]


OPENING_CASES = [   # name, game, limit, max_actions, ef, es, c_puct, alpha, seed, slot, seq, salt, opening_actions
    ("gmk_puct_open", "Gomoku", 30, 6, 6, 4, 4.5, 0.05, 77, 4, 0, 5, [[[7, 7], 0.333]]),           # Gomoku/Gomoku.py:49-50
    ("c4_puct_open_a", "Connect4", 30, 42, 8, 7, 2.5, 0.5, 78, 0, 0, 6, [[3, 0.5], [2, 0.2]]),
    ("c4_puct_open_b", "Connect4", 30, 42, 8, 7, 2.5, 0.5, 78, 1, 0, 6, [[3, 0.5], [2, 0.2]]),
    ("c4_puct_open_c", "Connect4", 30, 42, 8, 7, 2.5, 0.5, 78, 2, 0, 6, [[3, 0.5], [2, 0.2]]),
]


def main():
    os.makedirs(GOLD, exist_ok=True)
    only = set(sys.argv[1:])
    for name, *cfg, opening in OPENING_CASES:
        if only and name not in only:
            continue
        fx = ref_selfplay_puct(*cfg, opening=opening)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
        print(name, "T =", len(fx["actions"]), "first move", fx["actions"][0], "evals", fx["evaluator_calls"], flush=True)
    for name, game, it, max_actions, ef, es, cp, alpha, seed, slot, seq, blocks, net_seed, gumbel in NET_CASES:
        if only and name not in only:
            continue
        sess = NetSession(game, blocks, "linear" if gumbel else "softmax", net_seed)
        fx = ref_selfplay_puct(game, it, max_actions, ef, es, cp, alpha, seed, slot, seq, 0, session=sess, gumbel=gumbel)
        fx.update(sess.arrays(fx["states"].shape[1:]), net_blocks=blocks, net_seed=net_seed, is_gumbel=int(gumbel is not None))
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
        print(name, "T =", len(fx["actions"]), "stats", fx["game_stats"], "evals", fx["evaluator_calls"], "distinct states", len(fx["eval_value"]), flush=True)
    for name, *cfg in SINGLE_GUMBEL_CASES:
        if only and name not in only:
            continue
        fx = ref_single_gumbel(*cfg)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
        print(name, "T =", len(fx["actions"]), "evals", fx["evaluator_calls"], flush=True)
    for name, *cfg in SINGLE_CASES + TAU_CASES:
        if only and name not in only:
            continue
        fx = ref_single_tree_puct(*cfg)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
        print(name, "T =", len(fx["actions"]), "evals", fx["evaluator_calls"], flush=True)
    for name, game, it, max_actions, m, c_visit, c_scale, seed, slot, seq, salt, *rest in GUMBEL_CASES:
        if only and name not in only:
            continue
        fx = ref_selfplay_puct(game, it, max_actions, 0, 0, 0.0, 0.0, seed, slot, seq, salt, gumbel=(m, c_visit, c_scale, bool(rest and rest[0])))
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
        print(name, "T =", len(fx["actions"]), "stats", fx["game_stats"], "evals", fx["evaluator_calls"], flush=True)
    for name, *cfg in PUCT_CASES:
        if only and name not in only:
            continue
        fx = ref_selfplay_puct(*cfg)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
        print(name, "T =", len(fx["actions"]), "stats", fx["game_stats"], "evals", fx["evaluator_calls"], flush=True)


if __name__ == "__main__":
    main()
