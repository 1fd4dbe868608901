"""Fixture-generation harness: imports the *reference* (read-only, /root/reference) in THIS container
only, with stub modules for its absent dependencies and its np.random / np.argsort calls redirected to
the injected-noise stream defined in oracle/gaz_det.h.  Contains no reference source.  Nothing here is
used at run time by the product, the GPU tests, smoke() or bench.py — only by tools/gen_golden.py and the
optional container-only cross-checks in tests/ (skipped when /root/reference is absent).

What is stubbed and why:
  numba        -> njit = identity decorator (the reference has a non-jit fallback, MCTS.py:122-129);
                  uint8 array arguments are widened to int64 at the call boundary because numpy-2
                  promotion overflows on `uint8 + (-4)` (Gomoku.py:203) where Numba does not.
  onnxruntime, diskcache, h5py -> inert stand-ins (an in-memory h5py.File so Self_Play.play can append).
Injected noise: np.random.dirichlet / randint / choice / gumbel inside MCTS.py, MCTS_Gumbel.py and
Self_Play.py draw from oracle.gaz_oracle.{dirichlet,pick,uniform,gumbel}, keyed by (seed, slot, game_seq,
tree, event).  np.argsort inside MCTS.py is pinned to kind="stable" (tie rule, see gaz_puct.c header).
"""
import importlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gaz_oracle as O  # noqa: E402


def reference_available():
    return os.path.isdir(REF) and os.path.exists(os.path.join(REF, "MCTS.py"))


# ------------------------------------------------------------------ stub modules
def _widen(a):
    if isinstance(a, np.ndarray) and a.dtype == np.uint8:
        return a.astype(np.int64)
    if isinstance(a, np.generic) and a.dtype == np.uint8:
        return np.int64(a)
    return a


def _njit(*args, **kwargs):
    def deco(fn):
        def wrapped(*a, **k):
            with np.errstate(all="ignore"):
                return fn(*[_widen(x) for x in a], **{kk: _widen(v) for kk, v in k.items()})
        wrapped.__name__ = getattr(fn, "__name__", "fn")
        wrapped.__wrapped__ = fn
        return wrapped
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return deco(args[0])
    return deco


class _FakeDataset:
    def __init__(self, data):
        self.data = np.array(data)

    def __getitem__(self, i):
        return self.data[i]

    def __setitem__(self, i, v):
        self.data[i] = v


class _FakeH5File:
    STORE = {}

    def __init__(self, path, mode="r", **kw):
        self.path = path
        self.d = _FakeH5File.STORE.setdefault(path, {})

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def keys(self):
        return self.d.keys()

    def __getitem__(self, k):
        return self.d[k]

    def create_dataset(self, name, shape=None, maxshape=None, dtype=None, data=None, chunks=None, **kw):
        arr = np.zeros(shape, dtype) if data is None else np.array(data, dtype=dtype)
        self.d[name] = _FakeDataset(arr)
        return self.d[name]


def install_stubs():
    nb = types.ModuleType("numba")
    nb.njit = _njit
    nb.jit = _njit
    nb.prange = range
    nb.types = types.SimpleNamespace()
    ext = types.ModuleType("numba.extending")
    ext.is_jitted = lambda f: False
    nb.extending = ext
    sys.modules["numba"] = nb
    sys.modules["numba.extending"] = ext
    ort = types.ModuleType("onnxruntime")
    ort.InferenceSession = type("InferenceSession", (), {})
    ort.get_available_providers = lambda: ["CPUExecutionProvider"]
    sys.modules["onnxruntime"] = ort
    dc = types.ModuleType("diskcache")
    dc.Cache = type("Cache", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["diskcache"] = dc
    h5 = types.ModuleType("h5py")
    h5.File = _FakeH5File
    sys.modules["h5py"] = h5


# ------------------------------------------------------------------ injected noise
class Stream:
    """One (slot, game_seq, tree) stream; event advances once per replaced np.random call."""

    def __init__(self, seed, slot, game_seq, tree):
        self.seed, self.slot, self.game_seq, self.tree, self.event = seed, slot, game_seq, tree, 0

    def next_event(self):
        e = self.event
        self.event += 1
        return e


class Injector:
    def __init__(self, seed, slot=0, game_seq=0):
        self.seed, self.slot, self.game_seq = seed, slot, game_seq
        self.current = None          # Stream of the tree currently executing
        self.game_stream = Stream(seed, slot, game_seq, 2)
        self.n_trees = 0

    def new_tree_stream(self):
        s = Stream(self.seed, self.slot, self.game_seq, self.n_trees)
        self.n_trees += 1
        return s

    # replacements -----------------------------------------------------
    def dirichlet(self, alpha):
        s = self.current
        alpha = np.asarray(alpha, np.float64)
        assert np.all(alpha == alpha[0])
        return O.dirichlet(s.seed, s.slot, s.game_seq, s.tree, s.next_event(), float(alpha[0]), alpha.size)

    def randint(self, low=0, high=None, size=None):
        assert low == 0 and size is None
        s = self.current
        return O.pick(s.seed, s.slot, s.game_seq, s.tree, s.next_event(), int(high))

    def choice(self, a, size=None, replace=True, p=None):
        # legacy np.random.choice(p=...): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, side='right')
        s = self.current if self.current is not None else self.game_stream
        purpose = 2 if s.tree != 2 else 4
        u = O.uniform(s.seed, s.slot, s.game_seq, s.tree, s.next_event(), purpose)
        n = a if isinstance(a, (int, np.integer)) else len(a)
        p = np.array(p, dtype=np.float64)
        assert p.size == n
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        idx = int(cdf.searchsorted(u, side="right"))
        vals = np.arange(n) if isinstance(a, (int, np.integer)) else np.asarray(a)
        return np.array([vals[idx]])

    def gumbel(self, loc=0.0, scale=1.0, size=None):
        s = self.current
        n = int(np.prod(size))
        return loc + scale * O.gumbel(s.seed, s.slot, s.game_seq, s.tree, s.next_event(), n).reshape(size)


class _RandomProxy:
    def __init__(self, inj):
        self._inj = inj

    def __getattr__(self, name):
        if name in ("dirichlet", "randint", "choice", "gumbel"):
            return getattr(self._inj, name)
        if name == "seed":
            return lambda *a, **k: None
        raise AttributeError(f"np.random.{name} is not injected — add it to tools/ref_shim.py before using it")


class _NPProxy:
    """Stands in for the module-global `np` of one reference module."""

    def __init__(self, inj, stable_argsort):
        self.random = _RandomProxy(inj)
        self._stable = stable_argsort

    def argsort(self, a, *args, **kw):
        if self._stable and not args and not kw:
            return np.argsort(a, kind="stable")
        return np.argsort(a, *args, **kw)

    def __getattr__(self, name):
        return getattr(np, name)


_REF = {}


def load_reference():
    """Import the reference's MCTS / MCTS_Gumbel / Self_Play / game modules (once)."""
    if _REF:
        return _REF
    if not reference_available():
        raise RuntimeError("reference not present")
    install_stubs()
    for sub in ("", "Connect4", "Gomoku", "TicTacToe"):
        p = os.path.join(REF, sub)
        if p not in sys.path:
            sys.path.append(p)
    _REF["MCTS"] = importlib.import_module("MCTS")
    _REF["MCTS_Gumbel"] = importlib.import_module("MCTS_Gumbel")
    _REF["Self_Play"] = importlib.import_module("Self_Play")
    _REF["Connect4"] = importlib.import_module("Connect4")
    _REF["Gomoku"] = importlib.import_module("Gomoku")
    _REF["TicTacToe"] = importlib.import_module("Tictactoe")
    np.seterr(all="warn")   # MCTS.py:7 sets 'raise' process-wide; undo outside the reference
    _install_tree_hooks()
    return _REF


_INJ = {"inj": None}


def _install_tree_hooks():
    """Wrap MCTS / MCTS_Gumbel entry points so the injector knows which tree is drawing."""
    for modname, clsname in (("MCTS", "MCTS"), ("MCTS_Gumbel", "MCTS_Gumbel")):
        cls = getattr(_REF[modname], clsname)
        orig_init, orig_run, orig_prune = cls.__init__, cls.run, cls.prune_tree

        def init(self, *a, __o=orig_init, **k):
            inj = _INJ["inj"]
            self._gaz_stream = inj.new_tree_stream() if clsname_is_puct(self) else inj.gumbel_stream()
            prev, inj.current = inj.current, self._gaz_stream
            try:
                __o(self, *a, **k)
            finally:
                inj.current = prev

        def run(self, *a, __o=orig_run, **k):
            inj = _INJ["inj"]
            prev, inj.current = inj.current, self._gaz_stream
            try:
                with np.errstate(all="ignore"):
                    out = __o(self, *a, **k)
                rec = getattr(inj, "run_log", None)
                if rec is not None:
                    rec.append((self, out))
                return out
            finally:
                inj.current = prev

        def prune(self, *a, __o=orig_prune, **k):
            inj = _INJ["inj"]
            prev, inj.current = inj.current, self._gaz_stream
            try:
                return __o(self, *a, **k)
            finally:
                inj.current = prev

        cls.__init__, cls.run, cls.prune_tree = init, run, prune


def clsname_is_puct(obj):
    return type(obj).__name__ == "MCTS"


def _gumbel_stream(self):
    # MCTS_Gumbel rebuilds its tree every move (Self_Play.py:152-153): one continuing stream per game
    if not hasattr(self, "_gs"):
        self._gs = Stream(self.seed, self.slot, self.game_seq, 0)
    return self._gs


Injector.gumbel_stream = _gumbel_stream


def activate(seed, slot=0, game_seq=0, stable_argsort=True):
    """Point the reference's np.random at a fresh injected stream; returns the Injector."""
    ref = load_reference()
    inj = Injector(seed, slot, game_seq)
    inj.run_log = []
    _INJ["inj"] = inj
    for m in ("MCTS", "MCTS_Gumbel", "Self_Play"):
        ref[m].np = _NPProxy(inj, stable_argsort)
    return inj


# ------------------------------------------------------------------ sessions for the reference side
class HashSession:
    """session.run twin of the oracle's hash evaluator (MCTS.py:224-235 contract)."""

    def __init__(self, A, salt):
        self.A, self.salt, self.calls = A, salt, 0

    def run(self, output_names, input_feed, **kw):
        x = input_feed["inputs"]
        self.calls += 1
        pol, val = O.hash_eval(x[0].astype(np.int8), self.A, self.salt)
        return [pol.reshape(1, -1), np.array([[val]], np.float32)]


class FnSession:
    def __init__(self, fn):
        self.fn, self.calls = fn, 0

    def run(self, output_names, input_feed, **kw):
        self.calls += 1
        pol, val = self.fn(input_feed["inputs"][0].astype(np.int8))
        return [np.asarray(pol, np.float32).reshape(1, -1), np.array([[val]], np.float32)]
