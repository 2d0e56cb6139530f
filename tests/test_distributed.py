"""world_size-2 gloo test of the multi-GPU layout (one process per GPU, streams sharded, one
weight broadcast, max-over-ranks timing) — runs on CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from siammot_amd import parallel
    from siammot_amd.config import get_default_cfg
    from siammot_amd.emm import EMM
    from siammot_amd.track_utils import build_track_utils
    r, w, lr = parallel.init_distributed(backend="gloo")
    assert (r, w, lr) == (rank, world, rank) and parallel.is_distributed()
    cfg = get_default_cfg(channels=32)
    torch.manual_seed(1000 + rank)                     # ranks start with DIFFERENT weights
    emm = EMM(cfg, build_track_utils(cfg))
    with torch.no_grad():
        for p in emm.parameters():
            p.add_(torch.randn_like(p))
    before = torch.cat([p.detach().reshape(-1) for p in emm.parameters()]).clone()
    nbytes = parallel.broadcast_module(emm, src=0)
    after = torch.cat([p.detach().reshape(-1) for p in emm.parameters()])
    gathered = [torch.zeros_like(after) for _ in range(world)]
    dist.all_gather(gathered, after)
    same = all(torch.equal(g, gathered[0]) for g in gathered)
    t = parallel.max_over_ranks(1.0 + rank)            # slowest rank defines the job time
    per_rank = parallel.gather_floats(10.0 + rank)     # every rank's step time on every rank
    assert per_rank == [10.0 + r_ for r_ in range(world)]
    allowed = sorted(os.sched_getaffinity(0))
    mine = parallel.pin_rank_to_cores(rank, world)     # disjoint core slices per rank
    if len(allowed) >= world:
        per = len(allowed) // world
        assert mine == allowed[rank * per:(rank + 1) * per] and sorted(os.sched_getaffinity(0)) == mine
    parallel.barrier()
    q.put((rank, nbytes, same, bool(torch.equal(before, after)), t, parallel.shard_streams(8, rank, world)))
    dist.destroy_process_group()


def test_weight_broadcast_and_timing_reduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_param = sum(p.numel() for p in __import__("siammot_amd.emm", fromlist=["EMM"]).EMM(
        __import__("siammot_amd.config", fromlist=["x"]).get_default_cfg(channels=32),
        None).parameters())
    for rank, nbytes, same, unchanged, t, streams in res:
        assert nbytes == 4 * n_param
        assert same                                    # every rank holds rank 0's weights
        assert unchanged == (rank == 0)                # rank 0 kept its own, rank 1 was overwritten
        assert t == 2.0                                # max over ranks of (1.0, 2.0)
        assert streams == list(range(rank, 8, world))  # stream i -> rank i mod world


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from siammot_amd import parallel
    assert not parallel.is_distributed()
    assert parallel.max_over_ranks(3.5) == 3.5
    assert parallel.broadcast_module(torch.nn.Linear(2, 2)) == 0
    assert parallel.shard_streams(5, 0, 1) == [0, 1, 2, 3, 4]
    assert parallel.gather_floats(1.25) == [1.25]


def test_bench_self_launch_builds_a_torchrun_command(monkeypatch):
    """``python bench.py --gpus N`` must start its own ranks instead of exiting."""
    sys.path.insert(0, ROOT)
    import argparse
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert bench.self_launch(argparse.Namespace(gpus=4)) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bench_gpus2_plain_command_launches_its_own_ranks():
    """The plain driver form ``python bench.py --gpus 2 ...`` (no torchrun around it): bench.py re-executes itself
    under torch.distributed.run, the ranks rendezvous, broadcast the weights, barrier, reduce the time, and rank 0
    prints exactly one JSON line.  With two or more devices visible the backend is nccl (= RCCL) and every rank
    owns a GPU; on a one-GPU box the two ranks share the device and gloo stands in (the JSON says so)."""
    import json
    import subprocess
    n_dev = torch.cuda.device_count()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "10",
           "--prewarm-ms", "50", "--no-cpu-baseline", "--extra-streams", "0"]
    if n_dev < 2:
        cmd.append("--allow-shared-gpu")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SMOT_DIST_BACKEND")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %s" % res.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 100 and d["scaling"] == "weak" and d["value"] > 0
    par = d["config"]["parallelism"]
    assert "streams x2" in par and "world_size=2" in par and "weights broadcast once: 0 B" not in par
    if n_dev >= 2:
        assert "backend=nccl (RCCL)" in par and "SHARE DEVICES" not in par
    else:
        assert "backend=gloo" in par and "RANKS SHARE DEVICES" in par
    # rank 0's result was checked against the oracle and the reference-generated golden file
    assert d["parity"]["rerun_bitwise_equal_to_timed_result"] is True
    assert d["parity"]["vs_oracle_fp32"]["min_iou"] > 1 - 1e-3


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus_without_the_flag():
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SMOT_DIST_BACKEND")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                          "--prewarm-ms", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=env)
    assert res.returncode != 0 and "only 1 GPU(s) visible" in (res.stderr + res.stdout)


@pytest.mark.gpu
def test_rccl_process_group_runs_its_collectives_in_a_world_of_one():
    """The collectives the N-GPU path uses — one flat broadcast, a MAX all-reduce of the elapsed time, barriers — on
    the REAL backend (nccl = RCCL on ROCm) with the communicator bound to the device, in a world of one rank: what a
    one-GPU box can execute of the RCCL path (library load, communicator creation, kernels, teardown).  Fresh
    interpreter: a process group is process-global."""
    import subprocess
    code = (
        "import os, sys, socket, torch\n"
        "sys.path.insert(0, %r)\n"
        "from siammot_amd import parallel\n"
        "import torch.distributed as dist\n"
        "s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')\n"
        "torch.cuda.set_device(0)\n"
        "dev = torch.device('cuda', 0)\n"
        "rank, world, local = parallel.init_distributed(backend='nccl', device=dev, force=True)\n"
        "assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
        "flat = torch.arange(303495, dtype=torch.float32, device=dev)            # the head's parameter count\n"
        "dist.broadcast(flat, src=0)\n"
        "t = torch.tensor([1.25], dtype=torch.float64, device=dev)\n"
        "dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "dist.barrier()\n"
        "torch.cuda.synchronize()\n"
        "assert float(t.item()) == 1.25 and float(flat[-1].item()) == 303494.0\n"
        "dist.destroy_process_group()\n"
        "print('RCCL-OK')\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SMOT_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode == 0 and "RCCL-OK" in res.stdout, (res.stdout + res.stderr)[-3000:]
