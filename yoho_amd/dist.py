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


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # test hooks: YOHO_DIST_BACKEND=gloo runs the multi-rank host logic without RCCL, YOHO_FORCE_DEVICE=<n> puts every rank on one
    # device (a world-2 run of the dataset driver on a single-GPU box: tests/test_gpu_dropin.py)
    backend = os.environ.get("YOHO_DIST_BACKEND", backend)
    if os.environ.get("YOHO_FORCE_DEVICE") is not None:
        local = int(os.environ["YOHO_FORCE_DEVICE"])
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local if world > 1 else torch.cuda.current_device())
    return rank, world, local


def broadcast_state_dict(sd, spec, src=0, device=None):
    """Rank `src` holds `sd` (dict of ndarrays); every rank returns the identical dict.
    The float tensors are packed into ONE flat buffer -> a single broadcast (~15 MB per network)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sd
    fkeys = [(k, shp) for k, shp in spec if not k.endswith("num_batches_tracked")]
    n = int(sum(int(np.prod(shp)) for _, shp in fkeys))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local_results]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local_results)
    return out


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_ranks(x, device=None):
    """the value every rank holds, as a list in rank order (on every rank)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(x)]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(v.item()) for v in out]


def max_over_ranks(x, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
