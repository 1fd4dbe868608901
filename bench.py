#!/usr/bin/env python3
"""bench.py — self-play positions/sec (whole job) of the MI355X engine on BASELINE.json's metric/config.

A "step" = `--waves-per-step` simulation waves of the hot path over the resident batch of games (every wave =
one tree-kernel launch + one evaluator forward over all G leaves); games restart on device as they finish, so G
stays constant.  value = plies played by all ranks (delta game_stats[1]) / wall time of the K timed steps
(barrier + synchronize on both sides, max over ranks).  Inputs are device resident: nothing crosses PCIe in the
timed region except the per-step stats read-back after it ends.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)     # 8 x 400 waves = ~20 plies of every game at 200 sims/move (measure >= 20 plies per game)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="connect4", choices=["connect4", "gomoku", "gumbel"],
                    help="connect4 = BASELINE configs[1] (headline); gomoku = configs[3]; gumbel = configs[4]")
    ap.add_argument("--games", type=int, default=0, help="concurrent games per GPU (0 = the config's value)")
    ap.add_argument("--sims", type=int, default=0, help="iteration_limit of every MCTS.run (0 = the config's value)")
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--waves-per-step", type=int, default=400)
    ap.add_argument("--evaluator", default="resnet", choices=["resnet", "hash"])
    ap.add_argument("--max-tree-sims", type=int, default=0, help="evaluation-free simulations per game per wave (0 = library default)")
    ap.add_argument("--cache-leg", type=int, default=-1, help="log2 entries of the evaluation cache used by the extra with_eval_cache leg (0 = skip the leg; default 24, Gomoku 0: 5 %% hits there)")
    ap.add_argument("--eval-cache", type=int, default=0, help="log2 entries of the on-device evaluation cache (SURVEY 8f rank 3); 0 = off (the headline number is measured with it off: every request goes through the evaluator)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    return ap.parse_args()


def cpu_baseline(args, net):
    """The oracle (C restatement of the reference's Self_Play.play, one game at a time, one evaluator call per leaf —
    the reference's execution model) on the host cores of this box, evaluator = the same network in PyTorch fp32 on
    CPU.  Bounded sample of the same workload (Connect4, same sims/move, same net)."""
    import torch
    from oracle import gaz_oracle as O
    O.build()
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)

    def ev(state):
        with torch.no_grad():
            p, v = net(torch.from_numpy(state[None].copy()))
        return p[0].numpy(), float(v[0, 0])
    t0 = time.time(); positions = 0; games = 0; evals = 0
    while time.time() - t0 < args.cpu_baseline_seconds:
        # one bounded game: cap plies so a single call stays within the sample budget
        r = O.selfplay_game("Connect4", args.sims, 3, 8, 7, 2.5, 0.5, 1234, games, 0, evaluator=ev)
        positions += r["T"]; games += 1; evals += r["total_evals"]
    dt = time.time() - t0
    return dict(value=positions / dt, unit="positions/s", cores=cores, kind="port",
                sample=f"{games} games cut at 3 plies = {positions} positions, {evals} evaluator calls in {dt:.1f}s; "
                       f"oracle/ C restatement, sequential games, batch-1 PyTorch fp32 CPU evaluator ({cores} threads)")


CONFIGS = {   # game, games/GPU, sims/move, blocks, max_actions, explore first/second, c_puct, alpha, search, m
    "connect4": ("Connect4", 4096, 200, 6, 42, 8, 7, 2.5, 0.5, "puct", 0),
    "gomoku": ("Gomoku", 2048, 400, 10, 150, 6, 4, 4.5, 0.05, "puct", 0),
    "gumbel": ("Connect4", 8192, 32, 6, 42, 8, 7, 2.5, 0.5, "gumbel", 7),
}


def main():
    args = parse()
    game, dG, dS, dB, max_actions, ef, es, cpuct, alpha, search, gm = CONFIGS[args.config]
    args.games = args.games or dG; args.sims = args.sims or dS; args.blocks = args.blocks or dB
    if args.cache_leg < 0:
        args.cache_leg = 0 if args.config == "gomoku" else 24
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local) if torch.cuda.is_available() else None)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    torch.cuda.set_device(local)
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET, EVAL_HASH, SEARCH_GUMBEL, SEARCH_PUCT
    from grok_alpha_zero_amd.net import NETS, flops_per_position
    from grok_alpha_zero_amd.parallel import reduce_stats

    G = args.games
    gumbel = search == "gumbel"
    net = NETS[game](args.blocks, seed=0, policy_head="linear" if gumbel else "softmax").eval()
    eng = SelfPlayEngine(game, G, args.sims, max_actions, ef, es, cpuct, alpha, seed=1234, slot_offset=rank * G, device=local,
                         evaluator=EVAL_RESNET if args.evaluator == "resnet" else EVAL_HASH, net_blocks=args.blocks,
                         hash_salt=7, ring_capacity=0, search=SEARCH_GUMBEL if gumbel else SEARCH_PUCT, gumbel_m=gm,
                         c_visit=50.0, c_scale=1.0, policy_is_logits=gumbel, max_tree_sims_per_wave=args.max_tree_sims, eval_cache_log2=args.eval_cache)
    if args.evaluator == "resnet":
        eng.load_weights(net.export_engine_weights())

    def barrier():
        eng.synchronize(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"engine ready: {G} games, {args.sims} sims/move, evaluator={args.evaluator}")
    for i in range(args.warmup):
        eng.run_waves(args.waves_per_step)
        eng.synchronize()
        log(f"warmup step {i} done")
    barrier()
    s0 = eng.stats()
    eng.timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run_waves(args.waves_per_step)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt:.3f}s for {args.steps} steps")
    s1 = eng.stats()
    tm = eng.timing()
    eng_kernel = eng.dominant_kernel()
    eng.timing_reset(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    delta = np.array([int(s1["plies"] - s0["plies"]), int(s1["game_stats"][2] - s0["game_stats"][2]),
                      s1["evals"] - s0["evals"], s1["sims"] - s0["sims"], s1["cache_hits"] - s0["cache_hits"]], np.int64)
    total = reduce_stats(delta, world)                      # the one collective of the path: counters only
    positions, games, evals, sims, hits = (int(x) for x in total)

    if rank == 0:
        HWc = net.H * net.W
        fl = flops_per_position(args.blocks, H=net.H, W=net.W) if game == "Connect4" else dict(total=2.0 * HWc * 9 * 128 * 128 * 2 * args.blocks)
        roof = None
        kname, kflops = eng_kernel
        if args.evaluator == "resnet" and tm["n_dominant"] > 0 and kflops > 0:
            avg_ms = tm["ms_dominant"] / tm["n_dominant"]
            ach = kflops / (avg_ms * 1e-3) / 1e12
            traffic = None
            # per-launch HBM bytes of the dominant kernel from the PMC passes (tools/pmc_traffic.py; measured on this workload only)
            import glob
            for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
                tj = json.load(open(tf))
                if args.config == "connect4" and G == 4096 and tj.get("kernel_tag") and kname.startswith(tj["kernel_tag"]):
                    traffic = tj.get("bytes_per_launch")
            roof = dict(bound="mfma", achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS, traffic=traffic,
                        kernel=kname, flops_per_launch=kflops, avg_launch_us=avg_ms * 1e3, launches=int(tm["n_dominant"]))
        label = {"connect4": "Connect4 6x7", "gomoku": "Gomoku 15x15", "gumbel": "Connect4 6x7 Gumbel (m=7)"}[args.config]
        out = dict(metric="self-play positions/sec (whole node), Connect4 200 sims/move, 1/2/4/8 GPU",
                   value=positions / dt, unit="positions/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="bf16", data="synthetic",
                   config=dict(workload=f"{label}, {G} concurrent games/GPU, {args.sims} sims/move (MCTS.run iteration_limit), "
                                        f"{args.blocks}-block x128 ResNet bf16, {search} self-play, random-init weights",
                               games_per_gpu=G, sims_per_move=args.sims, evaluator=args.evaluator, waves_per_step=args.waves_per_step,
                               parallelism=f"games sharded x{world}, counters all-reduced"),
                   detail=dict(positions=positions, games_finished=games, evaluator_calls=evals, simulations=sims,
                               evals_per_position=evals / max(positions, 1), evals_per_s=evals / dt, sims_per_s=sims / dt,
                               eval_cache_log2=args.eval_cache, eval_cache_hits=hits,
                               eval_tflops=(evals - hits) * fl["total"] / dt / 1e12,
                               ms_tree_kernel_per_wave=tm["ms_tree"] / max(tm["n_waves"], 1),
                               ms_evaluator_per_wave=tm["ms_eval"] / max(tm["n_waves"], 1)),
                   roofline=roof)
        if not args.no_cpu_baseline and world == 1 and args.config == "connect4":
            out["cpu_baseline"] = cpu_baseline(args, net)
    eng.close()
    # ---- extra leg (not the headline): the same workload with the on-device evaluation cache (the reference's
    # Connect4 config runs its Session_Cache too, max_cache_depth = 2).  Search results are bit-identical; requests that repeat a
    # state already evaluated are answered from HBM inside the tree kernel and cost no wave.
    if args.cache_leg and args.eval_cache == 0 and args.evaluator == "resnet":
        eng2 = SelfPlayEngine(game, G, args.sims, max_actions, ef, es, cpuct, alpha, seed=1234, slot_offset=rank * G, device=local,
                              evaluator=EVAL_RESNET, net_blocks=args.blocks, hash_salt=7, ring_capacity=0,
                              search=SEARCH_GUMBEL if gumbel else SEARCH_PUCT, gumbel_m=gm, c_visit=50.0, c_scale=1.0,
                              policy_is_logits=gumbel, max_tree_sims_per_wave=args.max_tree_sims, eval_cache_log2=args.cache_leg)
        eng2.load_weights(net.export_engine_weights())
        for _ in range(args.warmup):
            eng2.run_waves(args.waves_per_step)
        eng2.synchronize(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        c0 = eng2.stats(); t0 = time.perf_counter()
        for _ in range(args.steps):
            eng2.run_waves(args.waves_per_step)
        eng2.synchronize(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt2 = time.perf_counter() - t0
        c1 = eng2.stats()
        if world > 1:
            tt = torch.tensor([dt2], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt2 = float(tt.item())
        d2 = reduce_stats(np.array([int(c1["plies"] - c0["plies"]), c1["evals"] - c0["evals"], c1["cache_hits"] - c0["cache_hits"]], np.int64), world)
        eng2.close()
        if rank == 0:
            out["with_eval_cache"] = dict(value=int(d2[0]) / dt2, unit="positions/s", entries_log2=args.cache_leg,
                                          hit_fraction=int(d2[2]) / max(int(d2[1]), 1), ms_per_step=dt2 / args.steps * 1e3,
                                          note="same search results bit for bit; NOT the headline value")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
