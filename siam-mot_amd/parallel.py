"""Multi-GPU layout of the EMM path: independent video streams, one process per GPU.

SURVEY.md §8(e): frame pairs of one stream are strictly sequential (track memory of frame t-1 feeds
frame t: reference rcnn.py:54,57), streams are independent, so the path shards by STREAM with no
per-frame collective.  The only exchange is a one-time broadcast of the weights from rank 0
(RCCL over xGMI on the GPU box; gloo in the CPU tests); track state stays per process.
The reference has no multi-GPU inference at all (README.md:70, engine/inferencer.py:156).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, device=None, force=False):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank).  No-op for a single process.  ``device`` (the rank's GPU, already made
    current by the caller) is handed to the RCCL process group so that its communicator binds to that device
    eagerly instead of guessing at the first collective.  ``force``: create the process group for a world of one as
    well (RCCL self-check on a one-GPU box: tests/test_distributed.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # SMOT_DIST_BACKEND=gloo: smoke-test the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("SMOT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


@torch.no_grad()
def broadcast_module(module, src=0):
    """Broadcast every parameter and buffer of ``module`` from rank ``src`` as ONE flat fp32
    buffer (a single collective: xGMI rings are per-link bound, so one large message beats many
    small ones).  Returns the number of bytes broadcast (0 when not distributed)."""
    # the tensors themselves, not ``.data`` aliases: an in-place copy through ``.data`` does not advance the
    # tensor's version counter, and derived parameter caches (ops.tower_packed) key on it
    tensors = list(module.parameters()) + list(module.buffers())
    tensors = [t for t in tensors if t.is_floating_point()]
    if not is_distributed() or not tensors:
        return 0
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
        off += n
    return flat.numel() * 4


def shard_streams(num_streams, rank, world):
    """Stream ids owned by ``rank``: stream i → rank i mod world."""
    return list(range(rank, num_streams, world))


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (elapsed time): the job is as slow as its slowest rank."""
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64,
                     device=device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if is_distributed():
        dist.barrier()


def gather_floats(value, device=None):
    """Every rank's ``value`` on every rank (a list of world_size floats): per-rank step times for the scaling line."""
    if not is_distributed():
        return [float(value)]
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    out = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(out, torch.tensor([float(value)], dtype=torch.float64, device=dev))
    return [float(t.item()) for t in out]


def pin_rank_to_cores(local_rank, local_world):
    """Give each rank of a node its own contiguous slice of the cores this process may run on (Linux numbers the cores
    of one socket / NUMA node contiguously, GPUs 0..3 / 4..7 of an MI355X node hang off sockets 0 / 1), so eight
    Python launch loops do not migrate across sockets or share a core.  Returns the slice (or None when the platform
    has no affinity call or the slice would be empty)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    per = len(cores) // max(local_world, 1)
    if per < 1:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


def shutdown():
    """Tear the process group down (each rank for itself, after its last collective): no complaint from the RCCL
    watchdog at interpreter exit when ranks finish at different times."""
    if is_distributed():
        dist.destroy_process_group()
