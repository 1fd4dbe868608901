"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libgaz_oracle.so (the CPU restatement of the reference's
PUCT search + Self_Play loop, see gaz_puct.c / gaz_selfplay.c).  Imported only by
tests/, tools/ (fixture generation), __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GAME_IDS = {"TicTacToe": 0, "Connect4": 1, "Gomoku": 2}
GAME_DIMS = {0: (3, 3, 2, 9), 1: (6, 7, 4, 7), 2: (15, 15, 2, 225)}  # H, W, C, A

EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))


class HashEvalCtx(C.Structure):
    _fields_ = [("salt", C.c_uint32), ("A", C.c_int)]


class SPConfig(C.Structure):
    _fields_ = [("game_id", C.c_int), ("run_iterations", C.c_int), ("max_actions", C.c_int),
                ("num_explore_actions_first", C.c_int), ("num_explore_actions_second", C.c_int),
                ("c_puct_init", C.c_double), ("c_puct_base", C.c_double), ("dirichlet_alpha", C.c_double),
                ("create_new_root", C.c_int), ("n_opening", C.c_int), ("opening_actions", C.c_int * 8), ("opening_weights", C.c_double * 8)]


class SPRecord(C.Structure):
    _fields_ = [("cap_T", C.c_int), ("T", C.c_int), ("winner", C.c_int),
                ("states", C.POINTER(C.c_int8)), ("policies", C.POINTER(C.c_float)),
                ("q", C.POINTER(C.c_float)), ("z", C.POINTER(C.c_float)), ("values", C.POINTER(C.c_float)),
                ("actions", C.POINTER(C.c_int)),
                ("root_N", C.POINTER(C.c_uint32)), ("root_W", C.POINTER(C.c_float)), ("root_P", C.POINTER(C.c_float)),
                ("root_visits", C.POINTER(C.c_uint64)), ("evals", C.POINTER(C.c_uint32)),
                ("total_evals", C.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "libgaz_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgaz_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgaz_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.gaz_api_dirichlet.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.POINTER(C.c_double)]
        L.gaz_api_pick.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.gaz_api_pick.restype = C.c_uint32
        L.gaz_api_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.gaz_api_uniform.restype = C.c_double
        L.gaz_api_gumbel.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
        L.gaz_api_log.argtypes = [C.c_double]; L.gaz_api_log.restype = C.c_double
        L.gaz_api_exp.argtypes = [C.c_double]; L.gaz_api_exp.restype = C.c_double
        L.gaz_api_np_sum_f32.argtypes = [C.POINTER(C.c_float), C.c_int]; L.gaz_api_np_sum_f32.restype = C.c_float
        L.gaz_api_np_sum_f64.argtypes = [C.POINTER(C.c_double), C.c_int]; L.gaz_api_np_sum_f64.restype = C.c_double
        L.gaz_api_philox.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.gaz_api_legal_actions.argtypes = [C.c_int, C.POINTER(C.c_int8), C.POINTER(C.c_int)]
        L.gaz_api_do_action.argtypes = [C.c_int, C.POINTER(C.c_int8), C.c_int, C.c_int]
        L.gaz_api_check_win.argtypes = [C.c_int, C.POINTER(C.c_int8), C.c_int, C.c_int]
        L.gaz_api_input_state.argtypes = [C.c_int, C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int8)]
        L.gaz_hash_eval.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.gaz_selfplay_game.argtypes = [C.POINTER(SPConfig), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(SPRecord)]
        L.gaz_selfplay_game.restype = C.c_int
        L.gaz_oracle_set_libm.argtypes = [C.c_int]
        L.gaz_selfplay_game_gumbel.argtypes = [C.POINTER(SPConfig), C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(SPRecord)]
        L.gaz_selfplay_game_gumbel.restype = C.c_int
        L.gaz_puct_best_index.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_int,
                                          C.c_uint64, C.c_double, C.c_double, C.c_int]
        L.gaz_puct_best_index.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---------------------------------------------------------------- injected noise stream
def dirichlet(seed, slot, seq, tree, event, alpha, n):
    out = np.empty(n, np.float64)
    lib().gaz_api_dirichlet(seed, slot, seq, tree, event, float(alpha), n, _p(out, C.c_double))
    return out


def pick(seed, slot, seq, tree, event, n):
    return int(lib().gaz_api_pick(seed, slot, seq, tree, event, n))


def uniform(seed, slot, seq, tree, event, purpose):
    return float(lib().gaz_api_uniform(seed, slot, seq, tree, event, purpose))


def gumbel(seed, slot, seq, tree, event, n):
    out = np.empty(n, np.float64)
    lib().gaz_api_gumbel(seed, slot, seq, tree, event, n, _p(out, C.c_double))
    return out


def np_sum_f32(a):
    a = np.ascontiguousarray(a, np.float32)
    return np.float32(lib().gaz_api_np_sum_f32(_p(a, C.c_float), a.size))


def np_sum_f64(a):
    a = np.ascontiguousarray(a, np.float64)
    return float(lib().gaz_api_np_sum_f64(_p(a, C.c_double), a.size))


def best_puct_index(priors, values, visits, parent_visits, c_init, c_base, use_libm=False):
    priors = np.ascontiguousarray(priors, np.float32); values = np.ascontiguousarray(values, np.float32)
    visits = np.ascontiguousarray(visits, np.uint32)
    return int(lib().gaz_puct_best_index(_p(priors, C.c_float), _p(values, C.c_float), _p(visits, C.c_uint32),
                                         priors.size, int(parent_visits), float(c_init), float(c_base), int(use_libm)))


# ---------------------------------------------------------------- hash evaluator (numpy twin of gaz_hash_eval)
def hash_eval_np(state_i8, A, salt):
    """Bit-exact numpy twin of gaz_hash_eval: state int8 array (any shape, C order)."""
    s = np.ascontiguousarray(state_i8, np.int8).reshape(-1).view(np.uint8)

    def h(seed):
        v = np.uint32(2166136261) ^ np.uint32(seed)
        with np.errstate(over="ignore"):
            for b in s:
                v = np.uint32((int(v) ^ int(b)) * 16777619 & 0xFFFFFFFF)
            v = int(v)
            v ^= v >> 16; v = (v * 0x85EBCA6B) & 0xFFFFFFFF; v ^= v >> 13; v = (v * 0xC2B2AE35) & 0xFFFFFFFF; v ^= v >> 16
        return v
    pol = np.empty(A, np.float32)
    for a in range(A):
        pol[a] = np.float32(((h(salt ^ (((a + 1) * 0x9E3779B1) & 0xFFFFFFFF)) >> 8) + 1)) * np.float32(2.0 ** -24)
    val = np.float32(h(salt ^ 0x51ED270B) >> 8) * np.float32(2.0 ** -23) - np.float32(1.0)
    return pol, np.float32(val)


def hash_eval(state_i8, A, salt):
    s = np.ascontiguousarray(state_i8, np.int8).reshape(-1)
    ctx = HashEvalCtx(salt, A)
    pol = np.empty(A, np.float32); val = C.c_float()
    lib().gaz_hash_eval(C.byref(ctx), _p(s, C.c_int8), s.size, _p(pol, C.c_float), C.byref(val))
    return pol, np.float32(val.value)


# ---------------------------------------------------------------- self-play of one game
def selfplay_game_gumbel(game, iteration_limit, max_actions, m, c_visit, c_scale, seed, slot=0, game_seq=0, evaluator=None,
                         hash_salt=0, use_libm=False, opening_actions=None, stablemax=False, gumbel_noise=True):
    """One Gumbel self-play game (MCTS_Gumbel.run(iteration_limit) per move, gumbel noise on).  stablemax=True: activation_fn =
    "stablemax" in the deterministic selection (build_config["use_stablemax"], Self_Play.py:69)."""
    return selfplay_game(game, iteration_limit, max_actions, 0, 0, 0.0, 0.0, seed, slot, game_seq, evaluator, hash_salt,
                         use_libm=use_libm, gumbel=(m, c_visit, c_scale, bool(stablemax), bool(gumbel_noise)), opening_actions=opening_actions)


def selfplay_game(game, run_iterations, max_actions, explore_first, explore_second, c_puct_init, dirichlet_alpha,
                  seed, slot=0, game_seq=0, evaluator=None, hash_salt=0, c_puct_base=19652.0, create_new_root=False,
                  use_libm=False, gumbel=None, opening_actions=None, live=None):
    """Play one PUCT self-play game with the oracle.  evaluator(state_i8[H,W,C]) -> (policy f32[A], value f32),
    or None for the built-in hash evaluator.  Returns a dict of numpy arrays (see gaz_sp_record)."""
    L = lib()
    gid = GAME_IDS[game] if isinstance(game, str) else int(game)
    H, W, Cc, A = GAME_DIMS[gid]
    cap = max_actions + 1
    rec = SPRecord(); rec.cap_T = cap
    if live is not None:
        live["rec"] = rec                              # rec.T = plies completed so far, readable from another thread while the game runs
    arrs = dict(states=np.zeros((cap, H, W, Cc), np.int8), policies=np.zeros((cap, A), np.float32),
                q=np.zeros(cap, np.float32), z=np.zeros(cap, np.float32), values=np.zeros(cap, np.float32),
                actions=np.zeros(cap, np.int32), root_N=np.zeros((cap, A), np.uint32), root_W=np.zeros((cap, A), np.float32),
                root_P=np.zeros((cap, A), np.float32), root_visits=np.zeros(cap, np.uint64), evals=np.zeros(cap, np.uint32))
    rec.states = _p(arrs["states"], C.c_int8); rec.policies = _p(arrs["policies"], C.c_float)
    rec.q = _p(arrs["q"], C.c_float); rec.z = _p(arrs["z"], C.c_float); rec.values = _p(arrs["values"], C.c_float)
    rec.actions = _p(arrs["actions"], C.c_int); rec.root_N = _p(arrs["root_N"], C.c_uint32)
    rec.root_W = _p(arrs["root_W"], C.c_float); rec.root_P = _p(arrs["root_P"], C.c_float)
    rec.root_visits = _p(arrs["root_visits"], C.c_uint64); rec.evals = _p(arrs["evals"], C.c_uint32)
    cfg = SPConfig(gid, run_iterations, max_actions, explore_first, explore_second, c_puct_init, c_puct_base,
                   dirichlet_alpha, int(create_new_root))
    for i, (a, w) in enumerate(opening_actions or []):     # [(action index, weight)], train_config["opening_actions"]
        cfg.opening_actions[i] = int(a); cfg.opening_weights[i] = float(w); cfg.n_opening = i + 1
    L.gaz_oracle_set_libm(int(use_libm))

    def play(fn, ctxp):
        if gumbel is None:
            L.gaz_selfplay_game(C.byref(cfg), fn, ctxp, seed, slot, game_seq, C.byref(rec))
        else:
            L.gaz_selfplay_game_gumbel(C.byref(cfg), int(gumbel[0]), float(gumbel[1]), float(gumbel[2]), run_iterations, fn, ctxp,
                                       seed, slot, game_seq, int(bool(use_libm)) | (2 if (len(gumbel) > 3 and gumbel[3]) else 0) | (4 if (len(gumbel) > 4 and not gumbel[4]) else 0),
                                       C.byref(rec))
    if evaluator is None:
        ctx = HashEvalCtx(hash_salt, A)
        play(C.cast(L.gaz_hash_eval, C.c_void_p), C.cast(C.byref(ctx), C.c_void_p))
    else:
        def cb(_ctx, state, n, pol, val):
            s = np.ctypeslib.as_array(state, shape=(n,)).reshape(H, W, Cc)
            p, v = evaluator(s)
            np.ctypeslib.as_array(pol, shape=(A,))[:] = np.asarray(p, np.float32).reshape(-1)
            val[0] = float(v)
        fn = EVAL_FN(cb)
        play(C.cast(fn, C.c_void_p), None)
    L.gaz_oracle_set_libm(0)
    T = rec.T
    out = {k: v[:T].copy() for k, v in arrs.items()}
    out["winner"] = rec.winner; out["T"] = T; out["total_evals"] = int(rec.total_evals)
    return out
