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
