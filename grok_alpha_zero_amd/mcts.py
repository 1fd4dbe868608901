"""MCTS / MCTS_Gumbel — the reference's search classes (MCTS.py:75-671, MCTS_Gumbel.py:151-733) as thin hosts of a
one-game engine: same constructor arguments, `run(iteration_limit, time_limit, use_bar) -> (move, rows)`,
`prune_tree(action, create_new_root)`, `update_hyperparams(**kw)`.  The tree, select / expand / backup, noise and move
sampling run in the HIP kernels; `session.run(["policy", "value"], {"inputs": x})` (MCTS.py:224-235) is called once per
simulation with a batch of one, exactly where the reference calls it.  `session=None` selects the synthetic evaluator
(the reference's uniform-random dummy, MCTS.py:237-241, is not reproducible by construction).

For throughput use `SelfPlayEngine` / `run_self_play` (thousands of games per launch); these classes exist so code written
against the reference (`<Game>/play.py`, `Game_Tester.py:480-513`) keeps working.
`time_limit` is honoured by `MCTS.run` (the host watches the clock, MCTS.py:560-563); `MCTS_Gumbel.run` maps it to the default
iteration budget like the reference (MCTS_Gumbel.py:573-578).  `use_njit` is ignored.
"""
import os
from warnings import warn

import numpy as np

from .engine import EVAL_EXTERNAL, EVAL_HASH, PH_HALT, PH_IDLE, PH_WAIT_HOST, SEARCH_GUMBEL, SEARCH_PUCT, SelfPlayEngine

_W = {"TicTacToe": 3, "Gomoku": 15}


def _game_name(game):
    return getattr(type(game), "ENGINE_NAME", type(game).__name__)


def _to_index(name, action):
    if name == "Connect4":
        return int(action)
    return int(action[1]) * _W[name] + int(action[0])


def _to_action(name, idx):
    if name == "Connect4":
        return np.int8(idx)
    dt = np.uint8 if name == "Gomoku" else np.int64
    return np.array([idx % _W[name], idx // _W[name]], dtype=dt)


class _EngineSearch:
    def _attach(self, game, session, seed, lib_path, **engine_kw):
        self.game, self.session = game, session
        self._name = _game_name(game)
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        self._eng = SelfPlayEngine(self._name, 1, 1, engine_kw.pop("max_actions"), 0, 0, engine_kw.pop("c_puct_init", 0.0),
                                   engine_kw.pop("dirichlet_alpha", 0.0), seed, sync_moves=True, single_tree=True,
                                   evaluator=EVAL_HASH if session is None else EVAL_EXTERNAL, ring_capacity=4,
                                   nodes_per_tree=32768,     # iteration limits are per run() call: size the arena generously
                                   lib_path=lib_path, **engine_kw)
        # the root (create_expand_root, MCTS.py:132) is built by the first run(): same evaluator call, same noise event
        self._eng.set_position(0, [_to_index(self._name, a) for a in game.action_history])

    def _pump(self, deadline=None):
        """Advance the engine until the game waits for the host; serve evaluator requests through session.run.  `deadline`
        (time.time() value): MCTS.run(time_limit=...) — when it passes, the search finishes its move at the next launch."""
        import time
        eng = self._eng
        eng.start_search()
        stopped = False
        try:
            if self.session is None and deadline is None:
                eng.run_move()
                return
            for _ in range(10_000_000):
                if deadline is not None and not stopped and time.time() >= deadline:
                    eng.stop_search(True); stopped = True
                if self.session is None:
                    eng.run_waves(4)
                    if eng.root_stats()["phase"][0] in (PH_WAIT_HOST, PH_HALT, PH_IDLE):
                        return
                    continue
                x, pend = eng.read_batch()
                if pend[0]:
                    policy, value = self.session.run(output_names=["policy", "value"],
                                                     input_feed={"inputs": np.expand_dims(x[0].astype(np.float32), 0)})
                    eng.write_outputs(np.asarray(policy, np.float32).reshape(1, -1), np.asarray(value, np.float32).reshape(-1))
                elif eng.root_stats()["phase"][0] in (PH_WAIT_HOST, PH_HALT, PH_IDLE):
                    return
                eng.wave_begin()
            raise RuntimeError("search did not finish")
        finally:
            if stopped:
                eng.stop_search(False)

    def _child_terminal(self, indices):
        """child.is_terminal of the root's children (MCTS.py:49,414,600): None while the game runs on, the winner (-1 / 1) or 0 for a
        draw when the move ends it — from the device's own rule code (probe_rules)."""
        hist = [_to_index(self._name, a) for a in self.game.action_history]
        w = self._eng.probe_rules([hist + [int(a)] for a in indices])["winner"] if len(indices) else []
        return {int(a): (None if int(x) == -2 else int(x)) for a, x in zip(indices, w)}

    def prune_tree(self, action, create_new_root=False):
        """game.do_action(action) was already called by the user (Self_Play.py:142-150); replay it on the device and re-root."""
        if create_new_root:
            self._eng.set_position(0, [_to_index(self._name, a) for a in self.game.action_history])
        else:
            self._eng.apply_moves([_to_index(self._name, action)])   # the new root's evaluation (if any) is requested by the next run()

    def close(self):
        self._eng.close()


class MCTS(_EngineSearch):
    _MAX_TIMED_ITERATIONS = 30_000               # one-game engines are created with a 32768-node arena per run() call

    def __init__(self, game, session=None, use_njit=None, c_puct_init=2.5, c_puct_base=19_652, use_dirichlet=True,
                 dirichlet_alpha=1.11, dirichlet_epsilon=0.25, tau=1.0, fast_find_win=False, *, seed=None, hash_salt=0,
                 max_actions=None, lib_path=None):
        self.c_puct_init, self.c_puct_base = c_puct_init, c_puct_base
        self.use_dirichlet, self.dirichlet_alpha, self.dirichlet_epsilon = use_dirichlet, dirichlet_alpha, dirichlet_epsilon
        self.tau = 0.0 if (tau != 0.0 and tau < 5e-3) else tau                          # MCTS.py:116-120
        self._attach(game, session, seed, lib_path, max_actions=max_actions or int(np.prod(game.board.shape)),
                     c_puct_init=c_puct_init, c_puct_base=c_puct_base, dirichlet_alpha=dirichlet_alpha,
                     dirichlet_epsilon=dirichlet_epsilon, use_dirichlet=use_dirichlet, hash_salt=hash_salt, search=SEARCH_PUCT,
                     fast_find_win=bool(fast_find_win))

    def update_hyperparams(self, **kwargs):
        """MCTS.py:134-168: invalid values are ignored with a warning, valid ones take effect at the next simulation."""
        upd = {}
        v = kwargs.get("c_puct_init")
        if v is not None:
            if v < 0.0: warn(f"c_puct_init value is invalid, {v} cannot be negative.")
            else: self.c_puct_init = upd["c_puct_init"] = v
        v = kwargs.get("c_puct_base")
        if v is not None:
            if v <= 0: warn("c_puct_base cannot be negative")
            else: self.c_puct_base = upd["c_puct_base"] = v
        v = kwargs.get("dirichlet_alpha")
        if v is not None:
            if v <= 0.0: warn("dirichlet_alpha cannot be less than or equal to 0")
            else: self.dirichlet_alpha = upd["dirichlet_alpha"] = v
        v = kwargs.get("dirichlet_epsilon")
        if v is not None:
            if v < 0.0 or v >= 1.0: warn("dirichlet_epsilon cannot be negative nor bigger than 1")
            else: self.dirichlet_epsilon = upd["dirichlet_epsilon"] = v
        tau = kwargs.get("tau")
        if tau is not None:
            if tau != 0.0 and tau <= 5e-3:
                warn("Tau can't be less than 5e-3. Changing tau = 0.0")
                tau = 0.0
            self.tau = tau                                                               # handed to the engine by the next run()
        if upd:
            self._eng.set_hyperparams(**upd)

    def run(self, iteration_limit=None, time_limit=None, use_bar=True):
        """-> (move, rows); row = [action, N / sum N, W / N, W, N, P, root.visits, is_terminal] sorted by visits (MCTS.py:591-618)."""
        import time
        n_legal = len(self.game.get_legal_actions())
        if time_limit is True:
            time_limit = 30.0                                                            # MCTS.py:547-548
        if n_legal == 1:
            iteration_limit = 1                                                          # MCTS.py:543-544
        elif time_limit is None and (iteration_limit is None or iteration_limit is True or iteration_limit < n_legal):
            iteration_limit = 3 * n_legal                                                # MCTS.py:545-546 (None would never stop there)
        if iteration_limit is None:
            iteration_limit = self._MAX_TIMED_ITERATIONS                                 # time-limited only: bounded by the node arena
        # tau = 0: most visited move; otherwise sample with weights N^(1/tau) in float64 (MCTS.py:602-612)
        self._eng.set_hyperparams(run_iterations=int(min(iteration_limit, self._MAX_TIMED_ITERATIONS)), tau=float(self.tau))
        self._pump(None if time_limit is None else time.time() + float(time_limit))
        st = self._eng.root_stats()
        N, Wv, P, rv = st["N"][0], st["W"][0], st["P"][0], int(st["root_visits"][0])
        idx = [a for a in np.argsort(-P, kind="stable") if N[a] > 0 or P[a] > 0]        # child order = descending prior
        total = float(N.sum())
        term = self._child_terminal(idx)
        rows = [[_to_action(self._name, a), N[a] / total, float(Wv[a]) / float(N[a]), Wv[a], N[a], P[a], rv, term[a]] for a in idx]
        rows.sort(key=lambda r: r[4], reverse=True)
        return _to_action(self._name, int(st["chosen"][0])), rows


class MCTS_Gumbel(_EngineSearch):
    def __init__(self, game, session, use_gumbel_noise=False, use_njit=None, m=16, c_visit=50.0, c_scale=0.1,
                 activation_fn="softmax", fast_find_win=False, *, seed=None, hash_salt=0, max_actions=None, lib_path=None):
        if activation_fn not in ("softmax", "stablemax"):
            raise ValueError("activation_fn must be 'softmax' or 'stablemax'")
        self.m, self.c_visit, self.c_scale, self.use_gumbel_noise = m, c_visit, c_scale, use_gumbel_noise
        self._attach(game, session, seed, lib_path, max_actions=max_actions or int(np.prod(game.board.shape)), hash_salt=hash_salt,
                     search=SEARCH_GUMBEL, gumbel_m=m, c_visit=c_visit, c_scale=c_scale, gumbel_stablemax=activation_fn == "stablemax",
                     fast_find_win=bool(fast_find_win), use_gumbel_noise=bool(use_gumbel_noise))

    def update_hyperparams(self, *args, **kwargs):                                       # MCTS_Gumbel.py:186-210
        upd = {k: kwargs[k] for k in ("m", "c_visit", "c_scale") if kwargs.get(k) is not None}
        if upd:
            self._eng.set_hyperparams(**upd)        # validates first (m >= 2, m / iterations within the node arena): a refused update leaves self untouched
        for k, v in upd.items():
            setattr(self, k, v)

    def run(self, iteration_limit=None, time_limit=None, use_bar=True):
        """-> (move, rows); row = [action, pi, mean value (pi where unvisited), W, N, logit, root.visits, is_terminal] sorted by pi."""
        if iteration_limit is None or iteration_limit is True or time_limit is not None:
            iteration_limit = 3 * len(self.game.get_legal_actions())                     # MCTS_Gumbel.py:573-578
        self._eng.set_search_params(int(iteration_limit), -1)
        self._pump()
        st = self._eng.root_stats()
        N, Wv, P, pi, rv = st["N"][0], st["W"][0], st["P"][0], st["policy"][0], int(st["root_visits"][0])
        legal = [_to_index(self._name, a) for a in self.game.get_legal_actions()]
        term = self._child_terminal(legal)
        rows = [[_to_action(self._name, a), pi[a], (float(Wv[a]) / float(N[a])) if N[a] else pi[a], Wv[a], N[a], P[a], rv, term[a]]
                for a in legal]
        rows.sort(key=lambda r: r[1], reverse=True)
        return _to_action(self._name, int(st["chosen"][0])), rows

    def prune_tree(self, action, create_new_root=False):
        # the reference rebuilds the Gumbel tree every move (Self_Play.py:151-153); so does the engine
        self._eng.apply_moves([_to_index(self._name, action)])
