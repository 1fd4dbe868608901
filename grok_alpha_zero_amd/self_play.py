"""run_self_play — the reference's self-play driver surface on top of the MI355X engine.

Mirrors `run_self_play(game_class, configs, folder_path, per_process_wait_time)` (Self_Play.py:259-413) and what
`Self_Play.play()` appends to the replay file (Self_Play.py:159-208): per finished game and per augmentation k,
datasets `boards_k` (board dtype), `policies_k` (f32), `values_k` (f32) with `values = 0.5 * (z + q)`, and the
`game_stats` u32[6] counters [max_len, total_actions, n_games, wins(-1), draws, wins(+1)].

Instead of N worker processes + a shared-memory inference server (Self_Play.py:334-400, Client_Server.py) all games run
concurrently on the GPU; `num_workers` therefore only bounds nothing here — the concurrency is `n_games`.

Storage: the reference writes HDF5 through h5py.  h5py is not installed in this image, so `ReplayStore` uses h5py when it is
importable and otherwise the system libhdf5 through h5io.py (ctypes) — a real `Self_Play_Data.h5` with the same dataset
names / dtypes / shapes / libver either way.
"""
import logging
import os

import numpy as np

from .engine import EVAL_HASH, EVAL_RESNET, SEARCH_GUMBEL, SEARCH_PUCT, SelfPlayEngine


class ReplayStore:
    """`Self_Play_Data.h5`: game_stats + boards_k / policies_k / values_k (Self_Play.py:178-208).  Backend: h5py when installed,
    else libhdf5 through h5io.py (a real HDF5 file either way), else — no HDF5 library at all — an .npy directory."""

    def __init__(self, folder_path):
        self.folder = folder_path
        try:
            import h5py  # noqa: F401
            self.backend = "h5py"
        except ImportError:
            from . import h5io
            self.backend = "libhdf5" if h5io.available() else "npy"
        self.path = os.path.join(folder_path, "Self_Play_Data.h5" if self.backend != "npy" else "Self_Play_Data.npzdir")
        self._buf, self._buf_bytes, self._in_session = [], 0, False

    def _open(self, mode):
        if self.backend == "h5py":
            import h5py
            return _H5pyAdapter(h5py.File(self.path, mode, libver="latest"))
        from .h5io import H5File
        return H5File(self.path, mode)

    def exists(self):
        return os.path.exists(self.path)

    def create(self):
        os.makedirs(self.folder, exist_ok=True)
        if self.backend == "npy":
            os.makedirs(self.path, exist_ok=True)
            np.save(os.path.join(self.path, "game_stats.npy"), np.zeros(6, np.uint32))
            return
        with self._open("w") as f:                                    # <Game>/main.py:83-86
            f.create_dataset("game_stats", np.zeros(6, np.uint32), maxshape=(6,), dtype=np.uint32)

    def game_stats(self):
        return self.read("game_stats")

    def n_datasets(self):
        if self.backend == "npy":
            return len([n for n in os.listdir(self.path) if n != "game_stats.npy"])
        with self._open_read() as f:
            return f.n_links() - 1

    # A generation appends thousands of games.  Opening the file and counting its datasets per game made the writer, not the GPU,
    # the limit (0.3 s per game, growing with the file); holding one handle open for the whole generation (round 1) left the
    # superblock's write flag set when the process was killed, and the file could not be reopened.  So `writing()` BUFFERS games and
    # writes them in batches — open, append `flush_every` games, close: the file is open for write only for the milliseconds of a
    # batch (the reference's exposure is one game's write, Self_Play.py:178-208), and a killed run keeps every batch it completed.
    def recover(self):
        """Repair a file that a KILLED writer left "open for write" (libhdf5 then refuses every open): clear the superblock's status flags, what
        `h5clear -s` does.  Only for the generation's single writer (run_self_play calls it before it reads the counters); a reader never
        repairs — the flag is also what a LIVE writer's open file looks like.  Returns True if a repair was made; any other reason for the
        file not opening is raised as it is."""
        if self.backend != "libhdf5" or not os.path.exists(self.path):
            return False
        try:
            self._open("r").close()
            return False
        except OSError as first:
            from .h5io import H5File, superblock_status_flags
            flags = superblock_status_flags(self.path)
            if not flags:                                             # not the interrupted-writer case: permissions, truncation, not HDF5 ...
                raise
            f = H5File(self.path, "r+", clear_status_flags=True)
            try:
                ok = f.n_links() >= 1 and f.read("game_stats").shape == (6,)      # must still be readable, else beyond this repair
            finally:
                f.close()
            if not ok:
                raise OSError(f"{self.path}: left inconsistent by an interrupted writer") from first
            logging.getLogger("grok_alpha_zero_amd").warning("%s was left open (status flags %#x) by an interrupted writer; flags cleared", self.path, flags)
            return True

    def _open_read(self):
        try:
            return self._open("r")
        except OSError as e:
            if self.backend == "libhdf5":
                from .h5io import superblock_status_flags
                if superblock_status_flags(self.path):
                    raise OSError(f"{self.path} is marked open for write: a writer is running, or one was killed — in that case the generation's "
                                  "writer repairs it (ReplayStore.recover(), which run_self_play calls); readers do not") from e
            raise

    def _open_rw(self):
        """the single writer's open: repairs after a killed predecessor"""
        try:
            return self._open("r+")
        except OSError:
            if not self.recover():
                raise
            return self._open("r+")

    def writing(self, flush_every=64, flush_bytes=64 << 20):
        store = self

        class _Session:
            def __enter__(self_inner):
                store._buf, store._buf_bytes = [], 0
                store._flush_every, store._flush_bytes, store._in_session = flush_every, flush_bytes, True
                return store

            def __exit__(self_inner, *a):
                try:
                    store._flush()
                finally:
                    store._in_session = False
                return False
        return _Session()

    def _flush(self):
        """append the buffered games / dataset triples / stats in ONE open-write-close"""
        buf, self._buf, self._buf_bytes = self._buf, [], 0
        if not buf:
            return
        if self.backend == "npy":
            for item in buf:
                self._write_npy(item)
            return
        f = self._open_rw()
        try:
            stats = f.read("game_stats").astype(np.uint32)
            k = (f.n_links() - 1) // 3                                # dataset_name = (len(keys) - 1) // 3 (Self_Play.py:190)
            if (f.n_links() - 1) % 3:                                 # a writer died inside a triple: drop its incomplete tail
                for nm in (f"boards_{k}", f"policies_{k}", f"values_{k}"):
                    if f.exists(nm):
                        f.delete(nm)
            for kind, payload in buf:
                if kind == "stats":
                    stats = payload.copy()
                    f.write("game_stats", stats)
                    continue
                if kind == "game":
                    boards_aug, policies_aug, values_aug, game_length, n_positions, winner = payload
                    stats[0] = max(int(stats[0]), game_length); stats[1] += n_positions; stats[2] += 1; stats[winner + 4] += 1
                    # the counters BEFORE the game's datasets, as the reference does (Self_Play.py:181-188 precede :190-208): a writer killed
                    # inside a batch then never leaves games in the file that game_stats[2] does not count — run_self_play resumes from that
                    # count (games_left, first_game_seq), and an under-count would replay the (slot, game_seq) streams of the uncounted games
                    f.write("game_stats", stats)
                    triples = [(boards_aug[i], policies_aug[i], values_aug[i]) for i in range(policies_aug.shape[0])]
                else:
                    triples = [payload]
                for b, p, v in triples:
                    f.create_dataset(f"boards_{k}", b, maxshape=(None, *b.shape[1:]), dtype=b.dtype)
                    f.create_dataset(f"policies_{k}", p, maxshape=(None, *p.shape[1:]), dtype=np.float32)
                    f.create_dataset(f"values_{k}", v, maxshape=(None, *v.shape[1:]), dtype=np.float32)
                    k += 1
        finally:
            f.close()

    def _write_npy(self, item):
        kind, payload = item
        if kind == "stats":
            np.save(os.path.join(self.path, "game_stats.npy"), payload)
            return
        k = self.n_datasets() // 3
        if kind == "game":
            boards_aug, policies_aug, values_aug, game_length, n_positions, winner = payload
            stats = self.game_stats().astype(np.uint32)
            stats[0] = max(int(stats[0]), game_length); stats[1] += n_positions; stats[2] += 1; stats[winner + 4] += 1
            np.save(os.path.join(self.path, "game_stats.npy"), stats)
            triples = [(boards_aug[i], policies_aug[i], values_aug[i]) for i in range(policies_aug.shape[0])]
        else:
            triples = [payload]
        for b, p, v in triples:
            np.save(os.path.join(self.path, f"boards_{k}.npy"), b)
            np.save(os.path.join(self.path, f"policies_{k}.npy"), np.asarray(p, np.float32))
            np.save(os.path.join(self.path, f"values_{k}.npy"), np.asarray(v, np.float32))
            k += 1

    def _put(self, item, nbytes):
        if not getattr(self, "_in_session", False):
            with self.writing():
                return self._put(item, nbytes)
        self._buf.append(item); self._buf_bytes += nbytes
        n_games = sum(1 for kind, _ in self._buf if kind != "stats")
        if n_games >= self._flush_every or self._buf_bytes >= self._flush_bytes:
            self._flush()

    def append_game(self, boards_aug, policies_aug, values_aug, game_length, n_positions, winner):
        """One finished game: arrays [n_aug, T, ...] (Self_Play.py:174-208)."""
        boards_aug = np.ascontiguousarray(boards_aug); policies_aug = np.ascontiguousarray(policies_aug, np.float32)
        values_aug = np.ascontiguousarray(values_aug, np.float32)
        self._put(("game", (boards_aug, policies_aug, values_aug, int(game_length), int(n_positions), int(winner))),
                  boards_aug.nbytes + policies_aug.nbytes + values_aug.nbytes)

    def append_datasets(self, boards, policies, values):
        """one (boards_k, policies_k, values_k) triple as the next k, without touching game_stats (shard merge, parallel.py)"""
        boards = np.ascontiguousarray(boards); policies = np.ascontiguousarray(policies, np.float32); values = np.ascontiguousarray(values, np.float32)
        self._put(("triple", (boards, policies, values)), boards.nbytes + policies.nbytes + values.nbytes)

    def set_game_stats(self, stats):
        self._put(("stats", np.asarray(stats).astype(np.uint32)), 24)

    def read(self, name):
        if self.backend == "npy":
            return np.load(os.path.join(self.path, name + ".npy"))
        with self._open_read() as f:
            return f.read(name)


class _H5pyAdapter:
    """h5py.File behind the four calls ReplayStore uses."""

    def __init__(self, f):
        self.f = f

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.f.close()
        return False

    def keys(self):
        return list(self.f.keys())

    def n_links(self):
        return len(self.f)

    def flush(self):
        self.f.flush()

    def exists(self, name):
        return name in self.f

    def delete(self, name):
        del self.f[name]

    def close(self):
        self.f.close()

    def create_dataset(self, name, data, maxshape=None, dtype=None):
        self.f.create_dataset(name, maxshape=maxshape, dtype=dtype, data=data, chunks=None)

    def write(self, name, data):
        self.f[name][...] = data

    def read(self, name):
        return np.array(self.f[name])


def _fast_states(name, actions):
    """Input states of every ply of one game for the built-in plugins (games.py), without a Python call per ply: [T, H, W, C]
    int8, equal to stacking game.get_input_state() before each move (checked against the plugin path in the tests)."""
    T = len(actions)
    if name == "Connect4":
        boards = np.zeros((T + 1, 6, 7), np.int8)                     # boards[t] = position before move t
        heights = [0] * 7
        player = -1
        for t, a in enumerate(actions):
            a = int(a)
            boards[t + 1] = boards[t]
            boards[t + 1, 5 - heights[a], a] = player
            heights[a] += 1; player = -player
        st = np.zeros((T, 6, 7, 4), np.int8)
        st[..., 3] = boards[:T]
        if T > 2: st[2:, :, :, 2] = boards[1:T - 1]                   # one move back, once two moves were played
        if T > 3: st[3:, :, :, 1] = boards[1:T - 2]
        cur = np.where(np.arange(T) % 2 == 0, 1, -1).astype(np.int8)  # -next_player: the first mover is -1
        st[..., 0] = cur[:, None, None]
        if T > 4: st[4:, :, :, 0] = boards[1:T - 3]                   # >= 4 moves: the board three moves back (Connect4.py:340-345)
        return st
    W = 3 if name == "TicTacToe" else 15
    boards = np.zeros((T + 1, W, W), np.int8)
    player = -1
    for t, a in enumerate(actions):
        a = int(a)
        boards[t + 1] = boards[t]
        boards[t + 1, a // W, a % W] = player
        player = -player
    st = np.empty((T, W, W, 2), np.int8)
    st[..., 1] = boards[:T]
    st[..., 0] = np.where(np.arange(T) % 2 == 0, -1, 1).astype(np.int8)[:, None, None]   # -current_player = next_player... see games.py:118
    return st


def record_to_samples(game_class, rec):
    """Rebuild what Self_Play.play() collected for one finished game (Self_Play.py:80,114-127,159-175): input states by
    replaying the moves through the Game plugin, improved policies, values = 0.5 (z + q), then augment_sample."""
    game = game_class()
    from . import games as _builtin
    if game_class in (_builtin.Connect4, _builtin.Gomoku, _builtin.TicTacToe):
        board_states = _fast_states(game_class.ENGINE_NAME, rec["actions"])
        n_plies = len(rec["actions"])
    else:
        states = []
        for idx in rec["actions"]:
            states.append(np.array(game.get_input_state()).copy())
            game.do_action(game_class.index_to_action(int(idx)) if hasattr(game_class, "index_to_action") else idx)
        board_states = np.array(states, dtype=game.board.dtype)
        n_plies = len(game.action_history)
    policies = np.asarray(rec["policies"], np.float32)
    values = np.asarray(rec["values"], np.float32).reshape(-1, 1)
    aug_b, aug_p = game.augment_sample(board_states, policies)
    aug_b, aug_p = np.asarray(aug_b), np.asarray(aug_p)
    aug_v = np.repeat(values[None], aug_p.shape[0], axis=0)
    return aug_b, aug_p, aug_v, n_plies


def run_self_play(game_class, configs, folder_path, per_process_wait_time=1e-3, *, n_games=1024, seed=None, weights=None,
                  device=0, slot_offset=0, hash_salt=0, lib_path=None, progress=None, eval_cache_log2=22, generation=None,
                  first_game_seq=None, allow_synthetic=False, engine_stats=None, game_groups=0):
    """Generate `games_per_generation - game_stats[2]` self-play games into `folder_path` (Self_Play.py:259-272).
    (`engine_stats`: a dict that receives the engine's counters — evaluator calls, simulations, waves — when the generation is done.
    `game_groups`: gaz_engine_config.game_groups, scheduling only: 0 = the library's choice, 1 = one batch.)
    `configs` = (build_config, train_config[, optimizer_config]).  `weights` = dict from net.export_engine_weights()
    (generation > 0); generation 0 (folder name "0") plays with the synthetic evaluator like the reference's
    session=None dummy (Self_Play.py:40, MCTS.py:237-241).

    Which games: the reference starts exactly the missing games and runs every one of them to its end (Self_Play.py:346-408).
    The engine restarts slots on device, so the admitted set is fixed up front the same way: slot g plays its k-th game iff
    k * G + g < games_left (`games_budget`), then halts; every admitted game is written, whatever order they finish in.  (Keeping
    the first games to FINISH instead would favour short games.)  RNG streams are keyed by (seed, slot, game_seq) and game_seq
    starts at `first_game_seq` (default: the games already in the file), so a resumed generation never replays a game even
    with the same seed."""
    build_config, train_config = configs[0], configs[1]
    store = ReplayStore(folder_path)
    if not store.exists():
        raise ValueError("Dataset file hasn't been created. Self play depends on that file!")     # Self_Play.py:264-265
    store.recover()                                 # this process is the generation's single writer: repair what a killed predecessor left
    games_done = int(store.game_stats()[2])
    games_left = int(train_config["games_per_generation"] - games_done)
    if games_left <= 0:
        return 0
    if generation is None:
        generation = int(str(folder_path).rstrip("/").split("/")[-1])                 # Self_Play.py:274
    name = getattr(game_class, "ENGINE_NAME", game_class.__name__)
    if generation > 0 and weights is None and not allow_synthetic:
        # the reference would fail loading model.onnx here; silently writing hash-evaluator games as training data is worse
        raise ValueError(f"generation {generation} needs network weights (weights=net.export_engine_weights()); "
                         "pass allow_synthetic=True to play with the synthetic evaluator on purpose")
    use_net = generation > 0 and weights is not None
    if use_net and name == "Connect4" and int(build_config.get("num_filters", 128)) != 128:
        from .engine import EngineError
        raise EngineError("the Connect4 trunk kernels are built for num_filters = 128 (Connect4/Build_Model.py's default); "
                          f"num_filters = {build_config.get('num_filters')} would need the block-0 projection path")
    G = min(n_games, games_left)
    if seed is None:
        seed = int.from_bytes(os.urandom(8), "little")                 # np.random.seed() from OS entropy (Self_Play.py:221)
    if first_game_seq is None:
        first_game_seq = games_done
    gumbel = bool(train_config.get("use_gumbel"))
    # PUCT runs int(1.5 * limit) iterations per move (Self_Play.py:99), Gumbel runs `limit` (Self_Play.py:110-112)
    iters = int(train_config["MCTS_iteration_limit"]) if gumbel else int(train_config["MCTS_iteration_limit"] * 1.5)
    eng = SelfPlayEngine(name, G, iters, train_config["max_actions"],
                         train_config.get("num_explore_actions_first", 0), train_config.get("num_explore_actions_second", 0),
                         train_config.get("c_puct_init", 0.0), train_config.get("dirichlet_alpha", 0.0), seed,
                         search=SEARCH_GUMBEL if gumbel else SEARCH_PUCT, gumbel_m=train_config.get("m", 0),
                         c_visit=train_config.get("c_visit", 50.0), c_scale=train_config.get("c_scale", 1.0),
                         # policy head: raw logits for Gumbel, else Stablemax or softmax (Build_Model.py:54-60)
                         policy_is_logits=1 if gumbel else (2 if build_config.get("use_stablemax") else 0),
                         gumbel_stablemax=bool(gumbel and build_config.get("use_stablemax")),
                         opening_actions=[(game_class.action_to_index(a) if hasattr(game_class, "action_to_index") else int(a), w)
                                          for a, w in train_config.get("opening_actions", []) or []],
                         create_new_root=train_config.get("create_new_root", False), slot_offset=slot_offset, device=device,
                         evaluator=EVAL_RESNET if use_net else EVAL_HASH, hash_salt=hash_salt,
                         net_blocks=build_config.get("num_resnet_layers", 0) if use_net else 0,
                         net_filters=build_config.get("num_filters", 128), ring_capacity=max(4 * G, 64), lib_path=lib_path,
                         eval_cache_log2=eval_cache_log2,    # on-device Session_Cache (Self_Play.py:234-236): same games, fewer waves
                         games_budget=games_left, first_game_seq=first_game_seq,
                         # Self_Play.py:35,100-112: every MCTS.run gets time_limit = MCTS_time_limit next to its iteration limit; the engine keeps a
                         # wall clock per game and move (PUCT) / runs 3 x legal moves iterations per move (Gumbel, MCTS_Gumbel.py:576-578)
                         move_time_limit=float(train_config.get("MCTS_time_limit") or 0.0), game_groups=game_groups)
    if train_config.get("MCTS_time_limit") and gumbel:
        logging.getLogger("grok_alpha_zero_amd").warning("Time limit isn't allowed for gumbel MCTS defaulting to use 3 * len_legal_actions")   # MCTS_Gumbel.py:578
    logging.getLogger("grok_alpha_zero_amd").info("run_self_play: generation %d, %d games on %d slots, evaluator = %s, game_seq from %d",
                                                  generation, games_left, G, "ResNet (HIP)" if use_net else "synthetic hash evaluator", first_game_seq)
    if use_net:
        eng.load_weights(weights)
    written = 0
    launch = G                                          # slots the launches cover (shrinks in the generation's tail, see repack)
    try:
        with store.writing():
            eng.run_waves(64)
            idle = 0
            while written < games_left:
                recs = eng.drain_finished()             # waits for the launches queued so far
                # tail of the generation: every game has been started and the slots halt one by one, but a wave still evaluates all
                # of them — once half of the covered slots are idle, move the live games together and shrink the launches
                remaining = games_left - written - len(recs)          # games not finished yet >= games still running
                if 0 < remaining and remaining * 2 <= launch and launch > 16:
                    _, launch = eng.repack()
                eng.run_waves(64)                       # queue the next ones right away: the GPU works while the host converts and writes
                for rec in recs:                        # every record is one of the admitted games (games_budget)
                    aug_b, aug_p, aug_v, length = record_to_samples(game_class, rec)
                    store.append_game(aug_b, aug_p, aug_v, length, rec["T"], rec["winner"])
                    written += 1
                    if progress:
                        progress(written, games_left)
                idle = 0 if recs else idle + 1
                if idle > 100000:
                    raise RuntimeError("self-play made no progress")
        if engine_stats is not None:
            engine_stats.update({k: v for k, v in eng.stats().items() if k != "game_stats"})
    finally:
        eng.close()
    return written
