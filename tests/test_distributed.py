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


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The multi-rank flow of bench.py (rendezvous, weight broadcast, barrier, max-over-ranks, one JSON line from
    rank 0) on a one-GPU box: two ranks on the same device with the gloo backend standing in for RCCL."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SMOT_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "100",
           "--warmup", "10", "--prewarm-ms", "50", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=root, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %s" % res.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 100 and d["scaling"] == "weak" and d["value"] > 0
    assert "streams x2" in d["config"]["parallelism"] and "weights broadcast once: 0 B" not in d["config"]["parallelism"]
