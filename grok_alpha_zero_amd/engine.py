"""ctypes binding of libgaz_engine.so (C ABI: include/gaz_engine.h) — the MI355X batched self-play engine.

The library is the hipcc build in this package directory; if it is missing the import of this module's
`load_library()` raises — there is no CPU fallback on the product path.  (tests/ may hand an explicit
`lib_path` to exercise the host logic against the one-lane CPU emulation build under tests/emu.)
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_PKG, "libgaz_engine.so")

GAME_IDS = {"TicTacToe": 0, "Connect4": 1, "Gomoku": 2}
GAME_DIMS = {0: (3, 3, 2, 9), 1: (6, 7, 4, 7), 2: (15, 15, 2, 225)}  # H, W, C, A
SEARCH_PUCT, SEARCH_GUMBEL = 0, 1
EVAL_HASH, EVAL_RESNET, EVAL_EXTERNAL = 0, 1, 2
PH_WAIT_HOST, PH_HALT, PH_IDLE = 5, 8, 9


ABI_VERSION = 4               # GAZ_ENGINE_ABI_VERSION of include/gaz_engine.h this binding was written against


class EngineConfig(C.Structure):       # gaz_engine_config — tests/test_abi.py checks names, order and sizeof against the header
    _fields_ = [("struct_size", C.c_uint32), ("game", C.c_int32), ("search", C.c_int32), ("n_games", C.c_int32), ("run_iterations", C.c_int32),
                ("max_actions", C.c_int32), ("num_explore_actions_first", C.c_int32), ("num_explore_actions_second", C.c_int32),
                ("c_puct_init", C.c_double), ("c_puct_base", C.c_double), ("dirichlet_alpha", C.c_double),
                ("dirichlet_epsilon", C.c_double), ("use_dirichlet", C.c_int32), ("create_new_root", C.c_int32),
                ("sync_moves", C.c_int32), ("nodes_per_tree", C.c_int32), ("ring_capacity", C.c_int32),
                ("seed", C.c_uint64), ("slot_offset", C.c_uint32), ("evaluator", C.c_int32), ("hash_salt", C.c_uint32),
                ("device", C.c_int32), ("net_blocks", C.c_int32), ("net_filters", C.c_int32), ("policy_is_logits", C.c_int32),
                ("gumbel_m", C.c_int32), ("c_visit", C.c_double), ("c_scale", C.c_double), ("compact_trees", C.c_int32),
                ("single_tree", C.c_int32), ("n_opening", C.c_int32), ("opening_actions", C.c_int32 * 8),
                ("opening_weights", C.c_double * 8), ("max_tree_sims_per_wave", C.c_int32), ("eval_cache_log2", C.c_int32), ("gumbel_stablemax", C.c_int32), ("fast_find_win", C.c_int32),
                ("no_gumbel_noise", C.c_int32), ("first_game_seq", C.c_uint32), ("games_budget", C.c_int64), ("tau", C.c_double), ("move_time_limit", C.c_double), ("game_groups", C.c_int32)]


class SearchHyperparams(C.Structure):  # gaz_search_hyperparams
    _fields_ = [("struct_size", C.c_uint32), ("use_dirichlet", C.c_int32), ("c_puct_init", C.c_double), ("c_puct_base", C.c_double),
                ("dirichlet_alpha", C.c_double), ("dirichlet_epsilon", C.c_double), ("tau", C.c_double), ("gumbel_m", C.c_int32),
                ("run_iterations", C.c_int32), ("c_visit", C.c_double), ("c_scale", C.c_double)]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class RecordLayout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("record_bytes", "max_T", "A", "t_pad", "off_hdr", "off_actions", "off_q",
                                         "off_root_visits", "off_evals", "off_policy", "off_N", "off_W", "off_P")]


_LIBS = {}


def load_library(lib_path=None):
    path = lib_path or os.environ.get("GAZ_ENGINE_LIB") or DEFAULT_LIB
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()'); "
                           "there is no CPU fallback")
    L = C.CDLL(path)
    H = C.c_void_p
    L.gaz_engine_abi_version.restype = C.c_int; L.gaz_engine_config_size.restype = C.c_int
    if L.gaz_engine_abi_version() != ABI_VERSION or L.gaz_engine_config_size() != C.sizeof(EngineConfig):
        raise RuntimeError(f"{path}: ABI version {L.gaz_engine_abi_version()} / config size {L.gaz_engine_config_size()} does not match "
                           f"this binding ({ABI_VERSION} / {C.sizeof(EngineConfig)}): rebuild the library")
    L.gaz_engine_create.argtypes = [C.POINTER(EngineConfig), C.POINTER(H)]
    L.gaz_engine_destroy.argtypes = [H]; L.gaz_engine_destroy.restype = None
    L.gaz_engine_last_error.argtypes = [H]; L.gaz_engine_last_error.restype = C.c_char_p
    L.gaz_engine_load_weights.argtypes = [H, C.POINTER(Tensor), C.c_int32]
    L.gaz_engine_reset_games.argtypes = [H, C.POINTER(C.c_int32), C.c_int32]
    L.gaz_engine_run_move.argtypes = [H, C.POINTER(C.c_int32)]
    L.gaz_engine_get_root_stats.argtypes = [H] + [C.c_void_p] * 8
    L.gaz_engine_apply_moves.argtypes = [H, C.POINTER(C.c_int32)]
    L.gaz_engine_run_waves.argtypes = [H, C.c_int32]
    L.gaz_engine_wave_begin.argtypes = [H]
    L.gaz_engine_wave_end.argtypes = [H]
    L.gaz_engine_batch_ptrs.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.gaz_engine_read_batch.argtypes = [H, C.c_void_p, C.c_void_p]
    L.gaz_engine_write_outputs.argtypes = [H, C.c_void_p, C.c_void_p]
    L.gaz_engine_evaluate.argtypes = [H, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
    L.gaz_engine_record_layout.argtypes = [H, C.POINTER(RecordLayout)]
    L.gaz_engine_drain_finished.argtypes = [H, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.gaz_engine_get_stats.argtypes = [H, C.POINTER(C.c_uint64)]
    L.gaz_engine_synchronize.argtypes = [H]
    L.gaz_engine_timing_reset.argtypes = [H, C.c_int32]
    L.gaz_engine_set_position.argtypes = [H, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    L.gaz_engine_set_search_params.argtypes = [H, C.c_int32, C.c_int32]
    L.gaz_engine_stop_search.argtypes = [H, C.c_int32]
    L.gaz_engine_start_search.argtypes = [H]
    L.gaz_engine_set_hyperparams.argtypes = [H, C.POINTER(SearchHyperparams)]
    L.gaz_engine_probe_rules.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 7
    L.gaz_engine_set_fused_wave.argtypes = [H, C.c_int32]
    L.gaz_engine_debug_fused_fault.argtypes = [H, C.c_int32]
    L.gaz_engine_read_positions.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32]
    L.gaz_engine_repack.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.gaz_engine_read_head_features.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.gaz_engine_dominant_kernel.argtypes = [H, C.c_char_p, C.c_int32, C.POINTER(C.c_double)]
    L.gaz_engine_timing_get.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    for f in ("create", "load_weights", "reset_games", "run_move", "get_root_stats", "apply_moves", "run_waves", "wave_begin",
              "wave_end", "batch_ptrs", "read_batch", "write_outputs", "evaluate", "record_layout", "drain_finished", "get_stats",
              "synchronize", "timing_reset", "timing_get", "dominant_kernel", "set_position", "set_search_params", "start_search", "stop_search",
              "set_hyperparams", "probe_rules", "read_head_features", "set_fused_wave", "debug_fused_fault", "read_positions", "repack"):
        getattr(L, "gaz_engine_" + f).restype = C.c_int
    _LIBS[path] = L
    return L


class EngineError(RuntimeError):
    pass


class SelfPlayEngine:
    """G concurrent self-play games on one GPU.  Mirrors, per game, what Self_Play(...).play() does in the
    reference (Self_Play.py:16-208) with MCTS.run / prune_tree underneath (MCTS.py:528-671)."""

    def __init__(self, game, n_games, run_iterations, max_actions, num_explore_actions_first, num_explore_actions_second,
                 c_puct_init, dirichlet_alpha, seed, *, c_puct_base=19652.0, dirichlet_epsilon=0.25, use_dirichlet=True,
                 create_new_root=False, sync_moves=False, nodes_per_tree=0, ring_capacity=None, slot_offset=0,
                 evaluator=EVAL_HASH, hash_salt=0, device=0, net_blocks=0, net_filters=128, search=SEARCH_PUCT,
                 policy_is_logits=False, max_tree_sims_per_wave=0, gumbel_m=0, c_visit=50.0, c_scale=1.0,
                 compact_trees=0, single_tree=False, opening_actions=None, eval_cache_log2=0, gumbel_stablemax=False, fast_find_win=False,
                 use_gumbel_noise=True, first_game_seq=0, games_budget=0, tau=-1.0, move_time_limit=0.0, game_groups=0, lib_path=None):
        self.L = load_library(lib_path)
        self.game_id = GAME_IDS[game] if isinstance(game, str) else int(game)
        self.H, self.W, self.Cc, self.A = GAME_DIMS[self.game_id]
        self.n_games = n_games
        if ring_capacity is None:
            ring_capacity = 2 * n_games
        self.cfg = EngineConfig(struct_size=C.sizeof(EngineConfig), game=self.game_id, search=search, n_games=n_games,
                                run_iterations=run_iterations, max_actions=max_actions, num_explore_actions_first=num_explore_actions_first,
                                num_explore_actions_second=num_explore_actions_second, c_puct_init=c_puct_init, c_puct_base=c_puct_base,
                                dirichlet_alpha=dirichlet_alpha, dirichlet_epsilon=dirichlet_epsilon, use_dirichlet=int(use_dirichlet),
                                create_new_root=int(create_new_root), sync_moves=int(sync_moves), nodes_per_tree=nodes_per_tree,
                                ring_capacity=ring_capacity, seed=seed, slot_offset=slot_offset, evaluator=evaluator, hash_salt=hash_salt,
                                device=device, net_blocks=net_blocks, net_filters=net_filters, policy_is_logits=int(policy_is_logits),
                                gumbel_m=gumbel_m, c_visit=c_visit, c_scale=c_scale, compact_trees=compact_trees, single_tree=int(single_tree),
                                max_tree_sims_per_wave=max_tree_sims_per_wave, eval_cache_log2=int(eval_cache_log2),
                                gumbel_stablemax=int(gumbel_stablemax), fast_find_win=int(fast_find_win),
                                no_gumbel_noise=int(not use_gumbel_noise), first_game_seq=int(first_game_seq), games_budget=int(games_budget),
                                tau=float(tau), move_time_limit=float(move_time_limit or 0.0), game_groups=int(game_groups))
        for i, (a, w) in enumerate(opening_actions or []):       # [(action index, weight)] — train_config["opening_actions"]
            self.cfg.opening_actions[i] = int(a); self.cfg.opening_weights[i] = float(w); self.cfg.n_opening = i + 1
        self.h = C.c_void_p()
        if self.L.gaz_engine_create(C.byref(self.cfg), C.byref(self.h)):
            raise EngineError(self.L.gaz_engine_last_error(None).decode())
        self.layout = RecordLayout()
        self._ck(self.L.gaz_engine_record_layout(self.h, C.byref(self.layout)))

    def _ck(self, rc):
        if rc:
            raise EngineError(self.L.gaz_engine_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.gaz_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ------------------------------------------------------------------------------------
    def load_weights(self, named_arrays):
        keep = []
        arr = (Tensor * len(named_arrays))()
        for i, (name, a) in enumerate(named_arrays.items()):
            a = np.ascontiguousarray(a, np.float32)
            keep.append(a)
            arr[i] = Tensor(name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size)
        self._ck(self.L.gaz_engine_load_weights(self.h, arr, len(named_arrays)))

    def evaluate(self, states_i8, repeats=0):
        """Evaluator probe (Compute_Speed.py:40-63): states int8 [n,H,W,C] -> (policy [n,A], value [n], ms per batch)."""
        x = np.ascontiguousarray(states_i8, np.int8)
        n = x.shape[0]
        pol = np.zeros((n, self.A), np.float32); val = np.zeros(n, np.float32); ms = C.c_double()
        self._ck(self.L.gaz_engine_evaluate(self.h, x.ctypes.data, n, pol.ctypes.data, val.ctypes.data, repeats, C.byref(ms)))
        return pol, val, ms.value

    def head_features(self, n):
        """Flat head features of the last evaluate() call (numerics tests): (p_feat [n, F], v_feat [n, F]) float32."""
        pr, vr = C.c_int32(), C.c_int32()
        self._ck(self.L.gaz_engine_read_head_features(self.h, 0, None, None, C.byref(pr), C.byref(vr)))
        p = np.zeros((n, pr.value), np.float32); v = np.zeros((n, vr.value), np.float32)
        self._ck(self.L.gaz_engine_read_head_features(self.h, n, p.ctypes.data, v.ctypes.data, None, None))
        return p, v

    # ---- synchronous per-move API ---------------------------------------------------------------------
    def run_move(self):
        n = C.c_int32()
        self._ck(self.L.gaz_engine_run_move(self.h, C.byref(n)))
        return n.value

    def root_stats(self):
        G, A = self.n_games, self.A
        out = dict(N=np.zeros((G, A), np.uint32), W=np.zeros((G, A), np.float32), P=np.zeros((G, A), np.float32),
                   policy=np.zeros((G, A), np.float32), root_visits=np.zeros(G, np.uint32), q=np.zeros(G, np.float32),
                   chosen=np.zeros(G, np.int32), phase=np.zeros(G, np.int32))
        self._ck(self.L.gaz_engine_get_root_stats(self.h, *[out[k].ctypes.data for k in
                                                           ("N", "W", "P", "policy", "root_visits", "q", "chosen", "phase")]))
        return out

    def apply_moves(self, moves=None):
        if moves is None:
            self._ck(self.L.gaz_engine_apply_moves(self.h, None))
        else:
            m = np.ascontiguousarray(moves, np.int32)
            self._ck(self.L.gaz_engine_apply_moves(self.h, m.ctypes.data_as(C.POINTER(C.c_int32))))

    def set_position(self, slot, action_indices):
        a = np.ascontiguousarray(action_indices, np.int32)
        self._ck(self.L.gaz_engine_set_position(self.h, int(slot), a.ctypes.data_as(C.POINTER(C.c_int32)), a.size))

    def read_positions(self):
        """-> list of action-index histories, one per slot: the game in progress (game.action_history; [] for a halted slot)"""
        T = int(self.layout.t_pad)
        n = np.zeros(self.cfg.n_games, np.int32); h = np.zeros((self.cfg.n_games, T), np.uint8)
        self._ck(self.L.gaz_engine_read_positions(self.h, n.ctypes.data, h.ctypes.data, T))
        return [h[g, :n[g]].astype(np.int32).tolist() for g in range(self.cfg.n_games)]

    def start_search(self):
        self._ck(self.L.gaz_engine_start_search(self.h))

    def stop_search(self, stop=True):
        self._ck(self.L.gaz_engine_stop_search(self.h, int(bool(stop))))

    def set_search_params(self, run_iterations=0, tau_mode=-1):
        self._ck(self.L.gaz_engine_set_search_params(self.h, int(run_iterations), int(tau_mode)))

    def set_hyperparams(self, *, c_puct_init=None, c_puct_base=None, dirichlet_alpha=None, dirichlet_epsilon=None, use_dirichlet=None,
                        tau=None, m=None, c_visit=None, c_scale=None, run_iterations=None):
        """MCTS.update_hyperparams / MCTS_Gumbel.update_hyperparams (MCTS.py:134-168, MCTS_Gumbel.py:186-210); None = unchanged."""
        nan = float("nan")
        d = lambda v: nan if v is None else float(v)
        hp = SearchHyperparams(struct_size=C.sizeof(SearchHyperparams), use_dirichlet=-1 if use_dirichlet is None else int(bool(use_dirichlet)),
                               c_puct_init=d(c_puct_init), c_puct_base=d(c_puct_base), dirichlet_alpha=d(dirichlet_alpha),
                               dirichlet_epsilon=d(dirichlet_epsilon), tau=d(tau), gumbel_m=-1 if m is None else int(m),
                               run_iterations=0 if run_iterations is None else int(run_iterations), c_visit=d(c_visit), c_scale=d(c_scale))
        self._ck(self.L.gaz_engine_set_hyperparams(self.h, C.byref(hp)))

    def probe_rules(self, histories, policy=None):
        """The device's game rules on a list of positions (each a list of action indices from the empty board): dict of board
        [n,H,W], legal [n,A] bool, winner [n], input [n,H,W,C], terminal [n,A] (-1 / 0 draw / 1 win) and, with `policy` [n,A],
        legal_policy [n,A] (get_legal_actions_policy_MCTS, normalize=True).  Guide.py:135-283, Game_Tester.py:297-405."""
        n = len(histories)
        stride = max(1, max((len(h) for h in histories), default=1))
        acts = np.zeros((n, stride), np.int32); na = np.zeros(n, np.int32)
        for i, h in enumerate(histories):
            na[i] = len(h); acts[i, :len(h)] = np.asarray(h, np.int32).reshape(-1)
        out = dict(board=np.zeros((n, self.H, self.W), np.int8), legal=np.zeros((n, self.A), np.uint8), winner=np.zeros(n, np.int32),
                   input=np.zeros((n, self.H, self.W, self.Cc), np.int8), terminal=np.zeros((n, self.A), np.int32))
        pin = pout = None
        if policy is not None:
            pin = np.ascontiguousarray(policy, np.float32); assert pin.shape == (n, self.A)
            pout = np.zeros((n, self.A), np.float32); out["legal_policy"] = pout
        self._ck(self.L.gaz_engine_probe_rules(self.h, acts.ctypes.data, na.ctypes.data, n, stride, out["board"].ctypes.data,
                                               out["legal"].ctypes.data, out["winner"].ctypes.data, out["input"].ctypes.data,
                                               out["terminal"].ctypes.data, None if pin is None else pin.ctypes.data,
                                               None if pout is None else pout.ctypes.data))
        out["legal"] = out["legal"].astype(bool)
        return out

    def reset_games(self, slots=None):
        if slots is None:
            self._ck(self.L.gaz_engine_reset_games(self.h, None, 0))
        else:
            s = np.ascontiguousarray(slots, np.int32)
            self._ck(self.L.gaz_engine_reset_games(self.h, s.ctypes.data_as(C.POINTER(C.c_int32)), s.size))

    # ---- continuous self-play -------------------------------------------------------------------------
    def run_waves(self, n):
        self._ck(self.L.gaz_engine_run_waves(self.h, int(n)))

    def synchronize(self):
        self._ck(self.L.gaz_engine_synchronize(self.h))

    # ---- external evaluator ---------------------------------------------------------------------------
    def wave_begin(self):
        self._ck(self.L.gaz_engine_wave_begin(self.h))

    def read_batch(self):
        x = np.zeros((self.n_games, self.H, self.W, self.Cc), np.int8)
        pend = np.zeros(self.n_games, np.int32)
        self._ck(self.L.gaz_engine_read_batch(self.h, x.ctypes.data, pend.ctypes.data))
        return x, pend

    def write_outputs(self, policy, value):
        p = np.ascontiguousarray(policy, np.float32); v = np.ascontiguousarray(value, np.float32)
        assert p.shape == (self.n_games, self.A) and v.size == self.n_games
        self._ck(self.L.gaz_engine_write_outputs(self.h, p.ctypes.data, v.ctypes.data))

    def batch_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(self.L.gaz_engine_batch_ptrs(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- results --------------------------------------------------------------------------------------
    def stats(self):
        out = (C.c_uint64 * 16)()
        self._ck(self.L.gaz_engine_get_stats(self.h, out))
        s = [int(x) for x in out]
        return dict(game_stats=np.array(s[:6], np.uint64), evals=s[6], sims=s[7], plies=s[8], waves=s[9], cache_hits=s[10], pipeline_groups=s[11], fused_wave=s[12], fused_faults=s[13], game_groups=max(int(s[14]), 1))

    def drain_finished(self, max_records=None):
        """Finished games as dicts: actions, policies [T,A], q, z, values (=0.5(z+q), Self_Play.py:165-172),
        root_N/W/P [T,A], root_visits, evals, winner, slot, game_seq."""
        lay = self.layout
        cap = max_records or max(self.cfg.ring_capacity, 1)
        buf = np.zeros((cap, lay.record_bytes), np.uint8)
        n = C.c_int32()
        self._ck(self.L.gaz_engine_drain_finished(self.h, buf.ctypes.data, cap, C.byref(n)))
        return [self.decode_record(buf[i]) for i in range(n.value)]

    def decode_record(self, raw):
        lay, A = self.layout, self.A
        hdr = raw[lay.off_hdr:lay.off_hdr + 16].view(np.int32)
        T, winner, slot, seq = (int(x) for x in hdr)

        def arr(off, dt, shape):
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            return raw[off:off + n].view(dt).reshape(shape).copy()
        actions = arr(lay.off_actions, np.uint8, (lay.t_pad,))[:T].astype(np.int32)
        q = arr(lay.off_q, np.float32, (lay.max_T,))[:T]
        mover = np.where(np.arange(T) % 2 == 0, -1.0, 1.0).astype(np.float32)   # target_z.append(next_player), Self_Play.py:127
        z = mover.copy()
        if winner == -1 and T and z[-1] == -1.0:
            z *= -1.0
        elif winner == 0:
            z[:] = 0.0
        return dict(T=T, winner=winner, slot=slot, game_seq=seq, actions=actions, q=q, z=z,
                    values=(np.float32(0.5) * (z + q)).astype(np.float32),
                    policies=arr(lay.off_policy, np.float32, (lay.max_T, A))[:T],
                    root_N=arr(lay.off_N, np.uint32, (lay.max_T, A))[:T], root_W=arr(lay.off_W, np.float32, (lay.max_T, A))[:T],
                    root_P=arr(lay.off_P, np.float32, (lay.max_T, A))[:T],
                    root_visits=arr(lay.off_root_visits, np.uint32, (lay.max_T,))[:T],
                    evals=arr(lay.off_evals, np.uint32, (lay.max_T,))[:T])

    # ---- measurement ----------------------------------------------------------------------------------
    def repack(self):
        """Move the games still running to the lowest slots and shrink the launches to them (generation tails); -> (active, launch size)."""
        a, b = C.c_int32(), C.c_int32()
        self._ck(self.L.gaz_engine_repack(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_fused_wave(self, on=True):
        """tree step + trunk kernel as one launch (default where available) or as separate launches; results do not change"""
        self._ck(self.L.gaz_engine_set_fused_wave(self.h, int(bool(on))))

    def debug_fused_fault(self, mod):
        """TEST HOOK: trunk workgroups with index % mod == 1 of the following fused launches give up their wait at once (0 = off)"""
        self._ck(self.L.gaz_engine_debug_fused_fault(self.h, int(mod)))

    def timing_reset(self, enable=True):
        self._ck(self.L.gaz_engine_timing_reset(self.h, int(enable)))

    def dominant_kernel(self):
        buf = C.create_string_buffer(1024); fl = C.c_double()
        self._ck(self.L.gaz_engine_dominant_kernel(self.h, buf, 1024, C.byref(fl)))
        return buf.value.decode(), fl.value

    def timing(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        d, e = C.c_int64(), C.c_int64()
        self._ck(self.L.gaz_engine_timing_get(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
        return dict(ms_tree=a.value, ms_eval=b.value, ms_dominant=c.value, n_dominant=d.value, n_waves=e.value)
