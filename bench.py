#!/usr/bin/env python3
"""bench.py — self-play positions/sec (whole job) of the MI355X engine on BASELINE.json's metric/config.

A "step" = `--waves-per-step` simulation waves of the hot path over the resident batch of games (every wave =
one tree-kernel launch + one evaluator forward over all G leaves); games restart on device as they finish, so G
stays constant.  value = plies played by all ranks (delta game_stats[1]) / wall time of the K timed steps
(barrier + synchronize on both sides, max over ranks).  Inputs are device resident: nothing crosses PCIe in the
timed region except the per-step stats read-back after it ends.

Multi-GPU: one process per GPU.  `python bench.py --gpus N` (no RANK in the environment) is the LAUNCHER: it touches no
GPU, checks that N devices exist, starts N fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
set, the fan-out of Self_Play.py:346-363), forwards rank 0's JSON line and fails if any rank fails.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already exist and each runs the bench body.
Games shard by global slot (rank r owns slots [r G, (r + 1) G)); the only collective is the all-reduce of the counters.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse(argv=None):
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
    ap.add_argument("--ref-convention-leg", type=int, default=-1, help="1: extra leg at int(1.5 * sims) simulations per move, what Self_Play passes for MCTS_iteration_limit = sims (Self_Play.py:99); default on for the connect4 config at N = 1")
    ap.add_argument("--other-configs", type=int, default=-1, help="1: also measure BASELINE configs[4] (Gumbel) and configs[3] (Gomoku) and append them to the line as "
                                                                  "'gumbel' / 'gomoku' objects (default: on for the plain headline command at N = 1)")
    ap.add_argument("--game-groups", type=int, default=0, help="gaz_engine_config::game_groups of the measured engines: 0 = the library's choice (2 for the headline config), "
                                                               "1 = one batch, K = K batches with their launches in flight together")
    ap.add_argument("--burn-in-waves", type=int, default=-1, help="untimed waves before the warm-up steps (default: about two game lengths for the Connect4 configs — 8000 PUCT / "
                                                                  "2400 Gumbel waves: from the lockstep start the ply mix of 4096 games settles to its stationary state within 16 x 400 waves, "
                                                                  "tools/ts_probe.py; 0 for Gomoku, whose games last minutes)")
    ap.add_argument("--stagger", type=int, default=-1, help="1: every slot's first game starts at a random ply of a random legal playout (default for Gomoku only: no burn-in can cover "
                                                            "a 150-ply game); 0: all games start at ply 0 (default for the Connect4 configs, which burn in instead)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-baseline-cores", type=int, default=0, help="worker threads = torch threads of the CPU baseline (0 = the cores this "
                                                                      "process may use: affinity, cut to the cgroup CPU quota)")
    ap.add_argument("--emu-lib", default="", help="TEST HOOK (tests/test_bench_launcher.py): run the bench body on the one-lane CPU emulation build of the "
                                                  "device code over gloo; never a measurement")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def visible_gpus():
    """Number of GPUs, WITHOUT initialising the runtime in this process (device_count() does not on this image): the launcher must
    stay GPU-free — a process that has touched the GPU may not be replaced or forked into ranks."""
    import torch
    return int(torch.cuda.device_count())


def launch(args, argv):
    """Parent of `python bench.py --gpus N`: N fresh children, one per GPU (Self_Play.py:346-363 starts one process per worker)."""
    n = args.gpus
    have = n if args.emu_lib else visible_gpus()
    if have < n:
        print(f"bench.py: --gpus {n} requested but {have} GPU(s) are visible: refusing to report a {n}-GPU number from fewer devices",
              file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().decode().splitlines()), daemon=True)
    reader.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in alive:
                    procs[q].terminate()                          # exact PIDs we started, never a pattern
        time.sleep(0.05)
    reader.join(timeout=10)
    lines = [ln for ln in out0 if ln.startswith("{")]
    if rc == 0 and not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr, flush=True)
        rc = 3
    if rc == 0:
        res = json.loads(lines[-1])
        if res.get("n_gpus") != n:
            print(f"bench.py: rank 0 reported n_gpus = {res.get('n_gpus')}, expected {n}", file=sys.stderr, flush=True)
            return 4
        print(lines[-1], flush=True)
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_cores():
    """CPU cores this process may really use: the scheduler affinity, cut down to the cgroup CPU quota when there is one (a GPU
    box exposes all of the host's logical CPUs to a container that is only granted a share of them)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                    # cgroup v2
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())   # v1
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n


def cpu_baseline(args, net):
    """The reference's execution model on this box's host cores: one game per worker, every leaf a blocking
    request to ONE inference server that batches whatever the workers have posted (Client_Server.py:162-217) — here W = all host
    cores worker threads, each playing oracle games (the C restatement of Self_Play.play; ctypes releases the GIL while it
    searches), and a server thread running the same network in PyTorch fp32 on the same cores.  Whole games, restarted as they
    finish; runs for --cpu-baseline-seconds and counts the plies completed in that window."""
    import queue
    import torch
    from oracle import gaz_oracle as O
    O.build()
    cores = args.cpu_baseline_cores or host_cores()
    W = cores
    torch.set_num_threads(cores)
    reqs = queue.Queue()
    stop = threading.Event()
    A = 7
    uniform = (np.full(A, 1.0 / A, np.float32), 0.0)
    stat = dict(batches=0, evals=0)

    def server():
        while True:
            first = reqs.get()
            if first is None:
                return
            batch = [first]
            t_end = time.perf_counter() + 3e-4                     # let the other workers' requests of this round arrive
            while len(batch) < W:
                try:
                    batch.append(reqs.get(timeout=max(0.0, t_end - time.perf_counter())))
                except queue.Empty:
                    break
                if batch[-1] is None:
                    batch.pop(); reqs.put(None); break
            if stop.is_set():
                for st, box, ev in batch:
                    box.append(uniform); ev.set()
                continue
            x = torch.from_numpy(np.stack([b[0] for b in batch]))
            with torch.no_grad():
                p, v = net(x)
            p = p.numpy(); v = v.numpy().reshape(-1)
            stat["batches"] += 1; stat["evals"] += len(batch)
            for i, (st, box, ev) in enumerate(batch):
                box.append((p[i], float(v[i]))); ev.set()

    workers = [dict(done=0, games=0, live={}) for _ in range(W)]

    def worker(w):
        def ev(state):
            if stop.is_set():
                return uniform                                     # drain: the window is over, finish the game without the network
            box, e = [], threading.Event()
            reqs.put((state.copy(), box, e)); e.wait()
            return box[0]
        k = 0
        while not stop.is_set():
            r = O.selfplay_game("Connect4", args.sims, 42, 8, 7, 2.5, 0.5, 1234, w, k, evaluator=ev, live=workers[w]["live"])
            if not stop.is_set():
                workers[w]["done"] += r["T"]; workers[w]["games"] += 1
            k += 1

    srv = threading.Thread(target=server, daemon=True); srv.start()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(W)]
    for t in th:
        t.start()
    time.sleep(args.cpu_baseline_seconds)
    # plies completed inside the window: finished games + the games in progress (rec.T is updated after every ply)
    positions = sum(wk["done"] + (int(wk["live"]["rec"].T) if "rec" in wk["live"] else 0) for wk in workers)
    dt = time.perf_counter() - t0
    stop.set()
    for t in th:
        t.join(timeout=120)
    reqs.put(None); srv.join(timeout=10)
    games = sum(wk["games"] for wk in workers)
    return dict(value=positions / dt, unit="positions/s", cores=cores, kind="port",
                sample=f"{W} concurrent whole games (one oracle game per host core, restarted as they finish) for {dt:.1f}s = {positions} plies, "
                       f"{games} games finished; {stat['evals']} evaluator requests served in {stat['batches']} batches (mean batch "
                       f"{stat['evals'] / max(stat['batches'], 1):.1f}) by one PyTorch fp32 CPU server on the same {cores} cores — the reference's "
                       f"worker + inference-server model (Self_Play.py:346-363, Client_Server.py:162-217) with oracle/ as the search",
                evals_per_s=stat["evals"] / dt, mean_batch=stat["evals"] / max(stat["batches"], 1))


CONFIGS = {   # game, games/GPU, sims/move, blocks, max_actions, explore first/second, c_puct, alpha, search, m
    "connect4": ("Connect4", 4096, 200, 6, 42, 8, 7, 2.5, 0.5, "puct", 0),
    "gomoku": ("Gomoku", 2048, 400, 10, 150, 6, 4, 4.5, 0.05, "puct", 0),
    "gumbel": ("Connect4", 8192, 32, 6, 42, 8, 7, 2.5, 0.5, "gumbel", 7),
}


# bytes one simulation moves through the tree kernel, SURVEY 8d: select d (12 A + 20) + backup 24 d + board 2 HW + new node 16 A' + 16 + bf16
# network input 2 HW C + outputs 4 (A_all + 1), at (A, d, HW, C) = (7, 8, 42, 4) for Connect4 and (200, 4, 225, 2) for Gomoku
TREE_BYTES_PER_SIM = {"Connect4": 8 * (12 * 7 + 20) + 24 * 8 + 2 * 42 + (16 * 7 + 16) + 42 * 4 * 2 + 4 * 8,
                      "Gomoku": 4 * (12 * 200 + 20) + 24 * 4 + 2 * 225 + (16 * 200 + 16) + 225 * 2 * 2 + 4 * 226}


def random_histories(game, n, rng, max_ply):
    """n random legal NON-TERMINAL action histories (engine action indices) of 0 .. max_ply plies: where bench.py puts the slots before
    the warm-up, so that the games are out of step from the start (a steady-state number) instead of all at ply 0."""
    from grok_alpha_zero_amd.games import GAMES
    cls = GAMES[game]
    out = []
    for _ in range(n):
        want = int(rng.integers(0, max_ply + 1))
        while True:
            g = cls(); hist = []; over = False
            for _ply in range(want):
                legal = g.get_legal_actions()
                a = legal[int(rng.integers(0, len(legal)))]
                g.do_action(a); hist.append(int(cls.action_to_index(a)))
                if g.check_win() != -2:                    # (check_win looks at the last action: never called on an empty board)
                    over = True
                    break
            if not over:
                break
        out.append(hist)
    return out


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch(args, argv))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: the launcher and the flag disagree", file=sys.stderr, flush=True)
        sys.exit(2)
    emu = bool(args.emu_lib)
    if not emu:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
        assert local < torch.cuda.device_count(), f"rank {rank}: no GPU {local} on this node"
        torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if emu else "nccl", rank=rank, world_size=world,
                                device_id=None if emu else torch.device("cuda", local))
    from grok_alpha_zero_amd.engine import SelfPlayEngine, EVAL_RESNET, EVAL_HASH, SEARCH_GUMBEL, SEARCH_PUCT
    from grok_alpha_zero_amd.net import NETS, flops_per_position
    from grok_alpha_zero_amd.parallel import reduce_stats

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def measure(config, G, sims_arg, blocks_arg, steps, warmup, legs):
        """One configuration, start to finish -> (the result dict on rank 0 / None elsewhere, the network)."""
        game, dG, dS, dB, max_actions, ef, es, cpuct, alpha, search, gm = CONFIGS[config]
        G = G or dG; sims = sims_arg or dS; blocks = blocks_arg or dB
        gumbel = search == "gumbel"
        use_net = args.evaluator == "resnet"
        net = NETS[game](blocks, seed=0, policy_head="linear" if gumbel else "softmax").eval() if (use_net or not args.no_cpu_baseline) else None
        weights = net.export_engine_weights() if use_net else None
        cache_leg = args.cache_leg if args.cache_leg >= 0 else (0 if config == "gomoku" else 24)
        # steady state.  The ply mix decides how many simulations of a move need the evaluator (openings ~190 of
        # 200, late plies a few dozen), i.e. positions/s.  From the lockstep start it oscillates with the period of a game length and is damped
        # out after ~16 x 400 waves (game lengths spread widely); drawing start positions "from the stationary distribution" by set_position was
        # tried four ways and each left a LARGER, slower-decaying oscillation (tools/ts_probe.py; gpurun_out r03 ts1 - ts9) — so: burn in.
        stagger = args.stagger if args.stagger >= 0 else int(game == "Gomoku")
        burn = args.burn_in_waves if args.burn_in_waves >= 0 else (0 if (game != "Connect4" or emu) else (2400 if gumbel else 8000))

        def make_engine(n_sims, cache_log2, groups=None):
            e = SelfPlayEngine(game, G, n_sims, max_actions, ef, es, cpuct, alpha, seed=1234, slot_offset=rank * G, device=0 if emu else local,
                               evaluator=EVAL_RESNET if use_net else EVAL_HASH, net_blocks=blocks if use_net else 0,
                               hash_salt=7, ring_capacity=0, search=SEARCH_GUMBEL if gumbel else SEARCH_PUCT, gumbel_m=gm,
                               c_visit=50.0, c_scale=1.0, policy_is_logits=gumbel, max_tree_sims_per_wave=args.max_tree_sims,
                               eval_cache_log2=cache_log2, game_groups=args.game_groups if groups is None else groups, lib_path=args.emu_lib or None)
            if use_net:
                e.load_weights(weights)
            if stagger:
                # Gomoku: a game is up to 150 plies of 400+ waves — minutes — so no burn-in reaches the stationary ply mix; every slot's first game
                # starts at a random ply (0 .. 60) of a seeded random legal playout instead, the same positions on every run
                hs = random_histories(game, G, np.random.default_rng(977 + rank), {"Gomoku": 60}.get(game, 4))
                for slot, h in enumerate(hs):
                    if h:
                        e.set_position(slot, h)
                e.synchronize()
            return e

        def barrier(e):
            e.synchronize()
            if not emu:
                torch.cuda.synchronize()
            if world > 1:
                dist.barrier()

        def max_over_ranks(dt):
            if world > 1:
                tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if emu else "cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt.item())
            return dt

        def timed_run(e, with_timing, burn_scale=1.0):
            """burn-in, W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; -> (seconds (max over ranks), counter deltas)"""
            for i in range(0, int(burn * burn_scale), 500):
                e.run_waves(min(500, int(burn * burn_scale) - i)); e.synchronize()
            for i in range(warmup):
                e.run_waves(args.waves_per_step); e.synchronize()
            barrier(e)
            s0 = e.stats()
            if with_timing:
                e.timing_reset(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                e.run_waves(args.waves_per_step)
            barrier(e)
            dt = time.perf_counter() - t0
            s1 = e.stats()
            d = np.array([int(s1["plies"] - s0["plies"]), int(s1["game_stats"][2] - s0["game_stats"][2]), s1["evals"] - s0["evals"],
                          s1["sims"] - s0["sims"], s1["cache_hits"] - s0["cache_hits"], s1["fused_faults"]], np.int64)
            return max_over_ranks(dt), d

        eng = make_engine(sims, args.eval_cache)
        log(f"[{config}] engine ready: {world} rank(s) x {G} games, {sims} sims/move, evaluator={args.evaluator}")
        dt, delta = timed_run(eng, True)
        log(f"[{config}] timed region done: {dt:.3f}s for {steps} steps")
        tm = eng.timing()
        kname, kflops = eng.dominant_kernel()
        fused = bool(eng.stats().get("fused_wave"))
        groups = int(eng.stats().get("game_groups", 1))
        tm_fused, kname_fused, tm_groups = None, None, None
        keng = eng                                              # the engine the per-kernel segments run on
        if groups > 1 and use_net:
            # The headline segment ran the games as `groups` batches with their launches in flight TOGETHER (gaz_engine_config::game_groups): the
            # durations of overlapping kernels price nothing.  The per-kernel numbers (roofline, roofline_tree, the fused launch) are therefore
            # taken on ONE batch of all the games — the launch shape the rocprofv3 summaries under profiles/ show — on a second engine brought to
            # the same steady state: same games, bit for bit.
            tm_groups = tm
            eng.timing_reset(False)
            keng = make_engine(sims, args.eval_cache, groups=1)
            for i in range(0, burn, 500):
                keng.run_waves(min(500, burn - i)); keng.synchronize()
            keng.timing_reset(True)
            keng.run_waves(args.waves_per_step); keng.synchronize()
            tm = keng.timing()
            kname, kflops = keng.dominant_kernel()
            log(f"[{config}] one-batch engine for the per-kernel segments ready")
        if fused and use_net:
            # The headline segment ran the tree step and the trunk kernel as ONE launch (k_wave_trunk): its duration includes the part of
            # the tree step it could not hide, so it does not price the MFMA kernel.  Time the trunk kernel on its own in a second
            # segment of this run: same engine, same games, tree step and trunk launched separately (results are bit-identical).
            tm_fused, kname_fused = tm, kname
            keng.set_fused_wave(False)
            keng.timing_reset(True)
            keng.run_waves(args.waves_per_step); keng.synchronize()
            tm = keng.timing()
            kname, kflops = keng.dominant_kernel()
            keng.set_fused_wave(True)
        keng.timing_reset(False)
        if keng is not eng:
            keng.close()
        t_red = time.perf_counter()
        total = reduce_stats(delta[:5], world)                  # the one collective of the path: counters only
        t_red = time.perf_counter() - t_red
        positions, games, evals, n_sims, hits = (int(x) for x in total)
        per_rank = np.zeros(world, np.int64); per_rank[rank] = delta[0]
        per_rank = reduce_stats(per_rank, world)
        faults = int(reduce_stats(delta[5:6], world)[0])
        eng.close()

        out = None
        if rank == 0:
            HWc = net.H * net.W if net is not None else 0
            fl = (flops_per_position(blocks, H=net.H, W=net.W) if game == "Connect4" else dict(total=2.0 * HWc * 9 * 128 * 128 * 2 * blocks)) if net is not None else dict(total=0.0)
            roof = None
            if use_net and tm["n_dominant"] > 0 and kflops > 0:
                avg_ms = tm["ms_dominant"] / tm["n_dominant"]
                ach = kflops / (avg_ms * 1e-3) / 1e12
                traffic, traffic_src = None, None
                # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes of this very command
                # (tools/profile_round.sh -> tools/pmc_traffic.py); a PMC pass cannot run inside the timed process, so the figure is
                # profile-derived and labelled as such.  Only quoted for the workload it was collected on.
                import glob
                for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_trunk_traffic.json"))):
                    tj = json.load(open(tf))
                    if config == "connect4" and G == 4096 and tj.get("kernel_tag") and kname.startswith(tj["kernel_tag"]):
                        traffic, traffic_src = tj.get("bytes_per_launch"), os.path.relpath(tf, ROOT)
                roof = dict(bound="mfma", achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS, traffic=traffic,
                            traffic_source=(f"{traffic_src}: FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes of this command, "
                                            "not measured in this run") if traffic_src else None,
                            kernel=kname, flops_per_launch=kflops, avg_launch_us=avg_ms * 1e3, launches=int(tm["n_dominant"]))
                import re
                m_issued = re.search(r"MFMA work issued = ([0-9.]+) of the counted FLOPs", kname)
                if m_issued:                                    # the kernel skips exact-zero work: say how much of the counted FLOPs the matrix cores execute
                    roof["mfma_work_issued_share"] = float(m_issued.group(1))
                    roof["frac_of_peak_issued"] = roof["frac"] * float(m_issued.group(1))
                    roof["issued_note"] = ("achieved / frac price the ALGORITHMIC FLOPs of the dense 3x3 convolutions (SURVEY 8d); the kernel leaves out the MFMA tiles whose "
                                           "16 cells all read zero padding on a tap (exact zeros, results bit-identical), so the matrix cores execute "
                                           "mfma_work_issued_share of them: frac_of_peak_issued is the hardware rate, frac the useful rate")
                if tm_fused is not None and tm_fused["n_dominant"] > 0:
                    f_ms = tm_fused["ms_dominant"] / tm_fused["n_dominant"]
                    roof["measured_in"] = (f"a second timed segment of this run ({args.waves_per_step} waves, HIP events on every 8th) with the tree step and the "
                                           "trunk launched as separate kernels; the headline segment runs them as ONE launch, see fused_launch" +
                                           ("; both on ONE batch of all the games (game_groups = 1, a second engine in the same steady state), see game_groups" if tm_groups else ""))
                    roof["fused_launch"] = dict(kernel=kname_fused, avg_launch_us=f_ms * 1e3, launches=int(tm_fused["n_dominant"]),
                                                frac_if_priced_as_mfma_only=kflops / (f_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                                                note="duration of tree step + trunk in one launch during the headline segment: the tree step's "
                                                     "slowest games (a latency-bound pointer chase) are partly hidden behind the trunk's first workgroups")
                if tm_groups is not None and tm_groups["n_dominant"] > 0:
                    g_ms = tm_groups["ms_dominant"] / tm_groups["n_dominant"]
                    wave_us = dt / max(steps * args.waves_per_step, 1) * 1e6
                    roof["game_groups"] = dict(groups=groups, launch_us_per_group=g_ms * 1e3, wave_us_all_groups=wave_us,
                                               trunk_kernel_alone_us=roof["avg_launch_us"], wave_over_trunk_alone=wave_us / roof["avg_launch_us"],
                                               note=f"headline segment: the games run as {groups} batches, each with its own stream and ONE launch (tree step + trunk) per wave, "
                                                    f"{groups} launches in flight together — a group's trunk tiles fill the chip while the other group's tree step starts and its heads "
                                                    "run.  launch_us_per_group is the duration of one group's launch while it shares the chip (overlapping: not a price); "
                                                    "wave_us_all_groups is wall clock per wave of all games; trunk_kernel_alone_us the MFMA kernel over all games on an otherwise idle chip")
                # the whole wave priced like the kernel: every row of every wave (tree step, heads, launch gaps and rows without a request included)
                wave_s = dt / max(steps * args.waves_per_step, 1)
                roof["whole_wave"] = dict(us=wave_s * 1e6, tflops=kflops / wave_s / 1e12, frac=kflops / wave_s / 1e12 / PEAK_BF16_TFLOPS,
                                          frac_useful_rows=evals / world * (kflops / G) / dt / 1e12 / PEAK_BF16_TFLOPS,
                                          note="flops_per_launch (all rows of one rank's batch) over the wall clock per wave of the headline segment: what the chip delivers end to "
                                               "end, tree step, heads and gaps included; frac_useful_rows counts only the rows that carried a request")
            # the tree kernel against ITS roofline (north_star: "rocprof HBM GB/s on tree kernels"): algorithmic bytes per launch = SURVEY 8d's bytes per
            # simulation x the simulations one launch runs (measured), over the launch duration of the separately-launched tree step (HIP events);
            # counter bytes from the committed --pmc passes.  It is a chain of dependent round trips, not a stream: the fraction says so.
            roof_tree = None
            tw = max(tm["n_waves"], 1)
            if tm["ms_tree"] > 0 and game in TREE_BYTES_PER_SIM:
                sims_per_launch = n_sims / world / max(steps * args.waves_per_step, 1)        # per rank = per launch
                t_us = tm["ms_tree"] / tw * 1e3
                alg = TREE_BYTES_PER_SIM[game] * sims_per_launch
                ctr, ctr_src = None, None
                import glob
                for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_tree_traffic.json"))):
                    tj = json.load(open(tf))
                    if config == "connect4" and G == 4096 and not gumbel and tj.get("bytes_per_launch"):
                        ctr, ctr_src = tj["bytes_per_launch"], os.path.relpath(tf, ROOT)
                exposed = None
                if tm_fused is not None and tm_fused["n_dominant"] > 0 and roof:
                    exposed = tm_fused["ms_dominant"] / tm_fused["n_dominant"] * 1e3 - roof["avg_launch_us"]
                roof_tree = dict(bound="hbm", kernel="k_wave_gumbel" if gumbel else ("k_wave_teams (four games per wavefront)" if game != "Gomoku" else "k_wave"),
                                 achieved=alg / (t_us * 1e-6) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=alg / (t_us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                                 algorithmic_bytes_per_launch=alg, bytes_per_simulation=TREE_BYTES_PER_SIM[game], simulations_per_launch=sims_per_launch,
                                 avg_launch_us=t_us, traffic=ctr, traffic_source=ctr_src,
                                 exposed_us_in_fused_launch=exposed,
                                 note="latency-bound pointer chase (one dependent L2 / HBM round trip per tree level): the launch lasts as long as its slowest game, "
                                      "so the HBM fraction is tiny by construction; what counts is how much of it the fused launch hides (exposed_us_in_fused_launch = "
                                      "fused launch - trunk kernel alone)")
            label = {"connect4": "Connect4 6x7", "gomoku": "Gomoku 15x15", "gumbel": "Connect4 6x7 Gumbel (m=7)"}[config]
            out = dict(metric="self-play positions/sec (whole node), Connect4 200 sims/move, 1/2/4/8 GPU",
                       value=positions / dt, unit="positions/s", n_gpus=world, steps=steps, warmup=warmup,
                       ms_per_step=dt / steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                       dtype="bf16", data="synthetic" if not emu else "synthetic — EMULATION BUILD ON CPU (launcher test), NOT A MEASUREMENT",
                       config=dict(workload=f"{label}, {G} concurrent games/GPU, {sims} sims/move (MCTS.run iteration_limit), "
                                            f"{blocks}-block x128 ResNet bf16, {search} self-play, random-init weights",
                                   games_per_gpu=G, sims_per_move=sims, evaluator=args.evaluator, waves_per_step=args.waves_per_step,
                                   start=("every slot's first game starts at a random ply (0 .. 60) of a seeded random legal playout (a Gomoku game lasts minutes: no burn-in reaches "
                                          "the stationary ply mix)" if stagger else f"all games at ply 0, then {burn} untimed burn-in waves + the warm-up steps: the ply mix of the "
                                          "games (which decides evaluations per position) has settled to its stationary state before the timed region"),
                                   parallelism=f"games sharded x{world} (rank r owns global slots [r G, (r + 1) G)), counters all-reduced"),
                       detail=dict(positions=positions, positions_per_rank=[int(x) for x in per_rank],
                                   positions_per_rank_min=int(per_rank.min()), positions_per_rank_max=int(per_rank.max()),
                                   counters_allreduce_ms=t_red * 1e3, games_finished=games,
                                   evaluator_calls=evals, simulations=n_sims,
                                   evals_per_position=evals / max(positions, 1), evals_per_s=evals / dt, sims_per_s=n_sims / dt,
                                   eval_cache_log2=args.eval_cache, eval_cache_hits=hits,
                                   eval_tflops=(evals - hits) * fl["total"] / dt / 1e12,
                                   fused_tree_and_trunk_launch=fused, fused_launch_faults=faults, game_groups=groups,
                                   ms_tree_kernel_per_wave=tm["ms_tree"] / tw,
                                   ms_evaluator_per_wave=tm["ms_eval"] / tw,
                                   per_wave_note=(("tree / evaluator ms per wave come from the unfused timing segment" + (" on one batch of all the games" if tm_groups else "")) if tm_fused is not None else None)),
                       roofline=roof, roofline_tree=roof_tree)

        # ---- extra legs (never the headline value) -------------------------------------------------------------------------
        # (1) the same workload with the on-device evaluation cache (the reference's Connect4 config runs its
        #     Session_Cache too).  Search results are bit-identical; repeated states are answered from HBM inside the tree kernel.
        if legs and cache_leg and args.eval_cache == 0 and use_net:
            e2 = make_engine(sims, cache_leg)
            dt2, d2 = timed_run(e2, False)
            t2 = reduce_stats(d2[:5], world)
            e2.close()
            if rank == 0:
                out["with_eval_cache"] = dict(value=int(t2[0]) / dt2, unit="positions/s", entries_log2=cache_leg,
                                              hit_fraction=int(t2[4]) / max(int(t2[2]), 1), ms_per_step=dt2 / steps * 1e3,
                                              note="same search results bit for bit; NOT the headline value")
        # (2) the reference's own convention: Self_Play passes int(1.5 * MCTS_iteration_limit) to MCTS.run (Self_Play.py:99), so a
        #     config that says "200" runs 300 simulations per move there (SURVEY 8d second line).
        ref_leg = args.ref_convention_leg if args.ref_convention_leg >= 0 else int(config == "connect4" and world == 1 and not gumbel)
        if legs and ref_leg and not gumbel:
            sims3 = int(sims * 1.5)
            e3 = make_engine(sims3, args.eval_cache)
            dt3, d3 = timed_run(e3, False, 1.5)
            t3 = reduce_stats(d3[:5], world)
            e3.close()
            if rank == 0:
                out["reference_convention"] = dict(value=int(t3[0]) / dt3, unit="positions/s", sims_per_move=sims3,
                                                   evals_per_position=int(t3[2]) / max(int(t3[0]), 1), ms_per_step=dt3 / steps * 1e3,
                                                   note=f"MCTS_iteration_limit = {sims} as Self_Play runs it: int(1.5 * limit) = {sims3} simulations per "
                                                        "move (Self_Play.py:99); NOT the headline value")
        return out, net

    out, net = measure(args.config, args.games, args.sims, args.blocks, args.steps, args.warmup, legs=True)
    # ---- BASELINE configs[3] and configs[4] in the same command: their own workloads, their own rooflines (priced per launch)
    # and evaluations per position; shorter timed regions (a Gomoku wave is six times a Connect4 wave).  Never the headline value.
    other = args.other_configs if args.other_configs >= 0 else int(args.config == "connect4" and world == 1 and not emu and args.games == 0 and args.sims == 0 and args.blocks == 0)
    if other:
        for name, steps, warm in (("gumbel", max(2, args.steps // 2), max(1, args.warmup // 2)), ("gomoku", max(2, args.steps // 4), 1)):
            o2, _ = measure(name, 0, 0, 0, steps, warm, legs=False)
            if rank == 0:
                keep = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "roofline_tree")
                out[name] = {k: o2[k] for k in keep}
                out[name]["evals_per_position"] = o2["detail"]["evals_per_position"]
                out[name]["evals_per_s"] = o2["detail"]["evals_per_s"]
                out[name]["ms_tree_kernel_per_wave"] = o2["detail"]["ms_tree_kernel_per_wave"]
                out[name]["ms_evaluator_per_wave"] = o2["detail"]["ms_evaluator_per_wave"]
                out[name]["note"] = f"BASELINE.json configs[{4 if name == 'gumbel' else 3}] measured by the same command; NOT the headline value"
    if rank == 0 and not args.no_cpu_baseline and world == 1 and args.config == "connect4" and not emu:
        args.sims = args.sims or CONFIGS["connect4"][2]
        out["cpu_baseline"] = cpu_baseline(args, net)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
