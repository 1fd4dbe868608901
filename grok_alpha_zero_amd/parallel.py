"""Multi-GPU: games shard by index across ranks (one process per GPU, no tree or replay state crosses GPUs,
Self_Play.py:346-363 already treats games as independent processes); the ONLY collective of the path is the
reduction of the small counter vector (game_stats, Self_Play.py:181-188) over RCCL (torch.distributed "nccl") — or
gloo in the CPU tests."""
import numpy as np


def shard_slots(n_games_per_rank, rank):
    """Global slot ids of this rank's games: RNG streams are keyed by global slot, so results do not depend on the GPU count."""
    return np.arange(rank * n_games_per_rank, (rank + 1) * n_games_per_rank, dtype=np.int64)


def reduce_stats(local_counts, world_size, max_fields=()):
    """Sum an int64 counter vector over ranks (fields listed in max_fields are max-reduced: game_stats[0] = longest game)."""
    local_counts = np.asarray(local_counts, np.int64)
    if world_size <= 1:
        return local_counts.copy()
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(local_counts.copy()).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = t.cpu().numpy()
    if max_fields:
        m = torch.from_numpy(local_counts.copy()).to(dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        mm = m.cpu().numpy()
        for f in max_fields:
            out[f] = mm[f]
    return out


def run_self_play_sharded(game_class, configs, folder_path, *, n_games=1024, seed=None, weights=None, lib_path=None, **kw):
    """`run_self_play` on every rank of an initialised torch.distributed job (one process per GPU; backend nccl = RCCL, or gloo):
    the generation's missing games are split over the ranks, each rank plays its share into a private shard file with RNG streams
    keyed by GLOBAL slot (rank r owns slots [r * n_games, (r + 1) * n_games)), then rank 0 appends the shards to
    `folder_path/Self_Play_Data.h5` in rank order.  The only collectives are the broadcast of the number of missing games and the
    all-reduce of the game_stats counters; no tree, evaluator or replay data crosses GPUs.  Returns the games written by all ranks.
    `seed=None`: rank 0 draws one from OS entropy (Self_Play.py:221) and every rank uses it; game sequence numbers start at the
    number of games already in the file, so a resumed generation adds new games even under a fixed seed."""
    import os
    import shutil
    import torch.distributed as dist
    from .self_play import ReplayStore, run_self_play
    rank, world = dist.get_rank(), dist.get_world_size()
    train_config = dict(configs[1])
    main = ReplayStore(folder_path)
    left = np.zeros(5, np.int64)                                       # [games missing, games done, the 64-bit seed in three 22-bit limbs]
    if rank == 0:
        if not main.exists():
            raise ValueError("Dataset file hasn't been created. Self play depends on that file!")
        main.recover()                                                 # rank 0 is the merged file's single writer
        done = int(main.game_stats()[2])
        left[0] = max(0, int(train_config["games_per_generation"]) - done); left[1] = done
        s0 = (int.from_bytes(os.urandom(8), "little") if seed is None else int(seed)) & 0xFFFFFFFFFFFFFFFF
        # every bit of the engine's 64-bit seed travels (limbs small enough for any integer reduction): a sharded run and a single-GPU run
        # given the same explicit seed use the same RNG streams
        left[2] = s0 & 0x3FFFFF; left[3] = (s0 >> 22) & 0x3FFFFF; left[4] = s0 >> 44
    left = reduce_stats(left, world)                                   # ranks > 0 contribute zeros: a broadcast through the one collective
    games_left, games_done = int(left[0]), int(left[1])
    seed = int(left[2]) | (int(left[3]) << 22) | (int(left[4]) << 44)
    share = games_left // world + (1 if rank < games_left % world else 0)
    generation = int(str(folder_path).rstrip("/").split("/")[-1])
    shard_dir = os.path.join(folder_path, f".shard{rank}")
    shutil.rmtree(shard_dir, ignore_errors=True)
    shard = ReplayStore(shard_dir); shard.create()
    device = kw.pop("device", int(os.environ.get("LOCAL_RANK", rank)) if dist.get_backend() == "nccl" else 0)
    if share > 0:
        run_self_play(game_class, (configs[0], dict(train_config, games_per_generation=share)) + tuple(configs[2:]), shard_dir,
                      n_games=min(n_games, share), seed=seed, weights=weights, device=device, slot_offset=rank * n_games,
                      lib_path=lib_path, generation=generation, first_game_seq=games_done, **kw)
    local = shard.game_stats().astype(np.int64)
    total = reduce_stats(local, world, max_fields=(0,))                # game_stats: [longest game (max), plies, games, wins -1, draws, wins +1]
    dist.barrier()
    if rank == 0:
        with main.writing():
            for r in range(world):
                sh = ReplayStore(os.path.join(folder_path, f".shard{r}"))
                n_sets = sh.n_datasets() // 3
                for k in range(n_sets):
                    main.append_datasets(sh.read(f"boards_{k}"), sh.read(f"policies_{k}"), sh.read(f"values_{k}"))
            before = main.game_stats().astype(np.int64)
            merged = before + total; merged[0] = max(int(before[0]), int(total[0]))
            main.set_game_stats(merged)
        for r in range(world):
            shutil.rmtree(os.path.join(folder_path, f".shard{r}"), ignore_errors=True)
    dist.barrier()
    return int(total[2])
