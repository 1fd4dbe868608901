"""CPU suite: bench.py's multi-GPU launch path.  `python bench.py --gpus N` must start N rank processes itself (the driver's N > 1
form when it does not go through torch.distributed.run), report n_gpus = N with the all-reduced counters, and refuse — loudly —
to report an N-GPU number from fewer devices.  The rank processes run the bench body on the one-lane emulation build of the
device code over gloo (`--emu-lib`, a test hook): the launcher, the sharding by global slot and the one collective are the real
code; the numbers are not measurements."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU = os.path.join(EMU_DIR, "libgaz_emu.so")
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--evaluator", "hash", "--games", "6", "--sims", "20", "--steps", "2", "--warmup", "1", "--waves-per-step", "40",
         "--no-cpu-baseline", "--cache-leg", "0", "--ref-convention-leg", "0"]


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return EMU


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}


def test_gpus_flag_fails_loudly_without_the_devices():
    """`--gpus N` was parsed and ignored, so a scaling run would have been N copies of N = 1.  On a box with fewer than
    N GPUs (this container has none) the launcher must exit non-zero and print no result line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box really has 2 GPUs")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(), capture_output=True, timeout=300)
    assert p.returncode != 0
    assert b"--gpus 2 requested" in p.stderr and b'"metric"' not in p.stdout


def test_gpus_2_launches_two_ranks_and_reports_the_aggregate(emu_lib):
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--emu-lib", emu_lib] + SMALL, env=_clean_env(), capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                          # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["steps"] == 2 and r["warmup"] == 1
    per = r["detail"]["positions_per_rank"]
    # round 3: the line diagnoses a multi-GPU run by itself — spread of the per-rank work and the wall time of the one collective
    assert r["detail"]["positions_per_rank_min"] == min(per) and r["detail"]["positions_per_rank_max"] == max(per) and r["detail"]["counters_allreduce_ms"] >= 0.0
    assert len(per) == 2 and all(x > 0 for x in per) and sum(per) == r["detail"]["positions"]
    assert abs(r["value"] - r["detail"]["positions"] / (r["ms_per_step"] * r["steps"] / 1e3)) < 1e-6 * r["value"]
    assert "x2" in r["config"]["parallelism"] and "NOT A MEASUREMENT" in r["data"]
    # the two ranks own different global slots (RNG streams keyed by global slot): their games differ
    assert per[0] != per[1] or r["detail"]["positions"] > 0
    # same flags, one rank: the aggregate of two ranks is about twice the work (weak scaling: per-rank work fixed)
    q = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--emu-lib", emu_lib] + SMALL, env=_clean_env(), capture_output=True, timeout=600)
    assert q.returncode == 0, q.stderr.decode()[-2000:]
    r1 = json.loads([ln for ln in q.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert r1["n_gpus"] == 1 and r1["detail"]["positions_per_rank"] == [per[0]]         # rank 0 of the 2-rank job played the same games
    assert r["detail"]["simulations"] > 1.5 * r1["detail"]["simulations"]


def test_rank_process_rejects_a_world_size_that_contradicts_the_flag(emu_lib):
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--emu-lib", emu_lib] + SMALL, env=env, capture_output=True, timeout=300)
    assert p.returncode == 2 and b"WORLD_SIZE" in p.stderr and b'"metric"' not in p.stdout


def test_launcher_propagates_a_failing_rank(tmp_path):
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--emu-lib", str(tmp_path / "missing.so")] + SMALL, env=_clean_env(),
                       capture_output=True, timeout=300)
    assert p.returncode != 0 and b'"metric"' not in p.stdout


def test_cpu_baseline_runs_workers_against_one_batching_server(oracle):
    """The bench's cpu_baseline leg: one oracle game per host core, every leaf a request to ONE batching PyTorch server
    (Client_Server.py:162-217) — here a short window with a one-block network."""
    sys.path.insert(0, ROOT)
    import bench
    from grok_alpha_zero_amd.net import Connect4Net
    args = bench.parse(["--sims", "12", "--cpu-baseline-seconds", "3"])
    r = bench.cpu_baseline(args, Connect4Net(1, seed=0).eval())
    assert r["kind"] == "port" and r["cores"] == bench.host_cores() <= len(os.sched_getaffinity(0)) and r["value"] > 0
    assert r["mean_batch"] > 1.0 or r["cores"] == 1                # requests of several workers really share a forward pass
    assert "whole games" in r["sample"]


def test_driver_form_torch_distributed_run(emu_lib):
    """The form the round-end driver uses for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` — the ranks already exist (RANK / WORLD_SIZE set by the launcher), bench.py must
    run the body in each and rank 0 print the one line with n_gpus = N."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--emu-lib", emu_lib] + SMALL
    p = subprocess.run(cmd, env=_clean_env(), capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and len(r["detail"]["positions_per_rank"]) == 2 and all(x > 0 for x in r["detail"]["positions_per_rank"])
