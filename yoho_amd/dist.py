"""Multi-GPU plumbing: one process per GPU, scene pairs shard with no data-path collective.

The reference has no distributed code at all; the path shards naturally
because PartI is independent per keypoint/fragment and matcher / Des2R / PartII / estimator are
independent per scene pair.  The only collective is a one-time broadcast of the
checkpoint tensors from rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" in CPU tests),
plus a host-side gather of the tiny per-pair results.
"""
import os
import numpy as np
import torch
import torch.distributed as dist

from . import weights as W


def _launched():
    """under torchrun / torch.distributed.run (or a test that sets the same variables): a rendezvous is described in the environment"""
    return "RANK" in os.environ and "MASTER_ADDR" in os.environ


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK / WORLD_SIZE / MASTER_*).

    The process group is created whenever the process was launched with a rendezvous in its environment - also at WORLD_SIZE = 1
    (`python -m torch.distributed.run --nproc-per-node 1 bench.py`): every collective below then really runs on the backend (RCCL
    for "nccl"), so the one-GPU box exercises the plumbing of the N-GPU run.  A plain `python bench.py` has no group and every
    helper returns the local value."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # test hooks: YOHO_DIST_BACKEND=gloo runs the multi-rank host logic without RCCL, YOHO_FORCE_DEVICE=<n> puts every rank on one
    # device (a world-2 run of the dataset driver on a single-GPU box: tests/test_gpu_dropin.py)
    backend = os.environ.get("YOHO_DIST_BACKEND", backend)
    forced = os.environ.get("YOHO_FORCE_DEVICE")
    if forced is not None:
        local = int(forced)
    if (world > 1 or _launched()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        # a device per rank is RCCL's contract; a gloo run stays device-agnostic (LOCAL_RANK may exceed the device count on a
        # single-GPU box) unless the caller pins it
        if torch.cuda.is_available() and (backend == "nccl" or forced is not None):
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available() and dist.is_initialized() and (dist.get_backend() == "nccl" or forced is not None):
        torch.cuda.set_device(local)
    return rank, world, local


def active():
    """a process group exists: the helpers below go through the backend (at any world size, 1 included)"""
    return dist.is_available() and dist.is_initialized()


def _coll_device(device=None):
    if device is not None:
        return device
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_state_dict(sd, spec, src=0, device=None):
    """Rank `src` holds `sd` (dict of ndarrays); every rank returns the identical dict.
    The float tensors are packed into ONE flat buffer -> a single broadcast (~15 MB per network)."""
    if not active():
        return sd
    fkeys = [(k, shp) for k, shp in spec if not k.endswith("num_batches_tracked")]
    n = int(sum(int(np.prod(shp)) for _, shp in fkeys))
    device = _coll_device(device)
    if dist.get_rank() == src:
        flat = np.concatenate([np.asarray(sd[k], dtype=np.float32).reshape(-1) for k, _ in fkeys])
        t = torch.from_numpy(flat).to(device)
    else:
        t = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(t, src=src)
    flat = t.cpu().numpy()
    out, o = {}, 0
    for k, shp in fkeys:
        c = int(np.prod(shp))
        out[k] = np.ascontiguousarray(flat[o:o + c].reshape(shp))
        o += c
    for k, shp in spec:
        if k.endswith("num_batches_tracked"):
            out[k] = np.zeros((), dtype=np.int64)
    return out


def shard(items, rank, world):
    """Round-robin shard of a list of independent work units (scenes, fragments, pairs)."""
    return list(items)[rank::world]


def gather_results(local_results):
    """Gather small python objects (per-pair (3,4) transforms etc.) on every rank, in rank order."""
    if not active():
        return [local_results]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local_results)
    return out


def barrier():
    if active():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def all_ranks(x, device=None):
    """the value every rank holds, as a list in rank order (on every rank)"""
    if not active():
        return [float(x)]
    device = _coll_device(device)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(v.item()) for v in out]


def max_over_ranks(x, device=None):
    if not active():
        return float(x)
    device = _coll_device(device)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
