"""RCCL on one GPU (-m gpu): the collectives of the N-GPU run, initialised and exercised at world size 1.

A single-GPU box cannot show a scaling curve, but it can show that nothing between `torch.distributed.run` and the first timed step
is dead code: the "nccl" (= RCCL) process group comes up, the checkpoint broadcast takes its DEVICE branch (one flat fp32 buffer on
cuda:0 through ncclBroadcast), the barrier / max-over-ranks / per-rank gather of the timed region and the host-side result gather
go through the backend, and `bench.py --gpus 1` runs end to end under the launcher the driver uses for N > 1.  The reference has
no distributed code on this path (its only collective is MinkowskiEngine/examples/multigpu_ddp.py:82-119); the contract tested
here is SURVEY 8(e): one broadcast of weights, no data-path collective, results gathered on the host.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "YOHO_DIST_BACKEND", "YOHO_FORCE_DEVICE"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    env.update({k: str(v) for k, v in kw.items()})
    return env


_WORKER = r"""
import json, sys
import numpy as np, torch, torch.distributed as dist
from yoho_amd import dist as ydist, weights as W
rank, world, local = ydist.init_from_env("nccl")
assert (rank, world, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl" and ydist.active()
assert torch.cuda.current_device() == 0
sd = W.synth_state_dict(W.PARTI_SPEC, 7)
got = ydist.broadcast_state_dict(sd, W.PARTI_SPEC)            # device branch: flat fp32 buffer on cuda:0 -> ncclBroadcast -> host
assert got is not sd and set(got) == set(sd)
for k, v in sd.items():
    if not k.endswith("num_batches_tracked"):
        assert got[k].dtype == np.float32 and np.array_equal(got[k], np.asarray(v, np.float32)), k
sd2 = W.synth_state_dict(W.PARTII_SPEC, 8)
got2 = ydist.broadcast_state_dict(sd2, W.PARTII_SPEC)
assert all(np.array_equal(got2[k], np.asarray(v, np.float32)) for k, v in sd2.items() if not k.endswith("num_batches_tracked"))
ydist.barrier()
assert ydist.all_ranks(1.25) == [1.25] and ydist.max_over_ranks(3.5) == 3.5
res = {"0-1": {"trans": np.arange(12.0).reshape(3, 4), "recalltime": 7}}
g = ydist.gather_results(res)
assert len(g) == 1 and np.array_equal(g[0]["0-1"]["trans"], res["0-1"]["trans"]) and g[0]["0-1"]["recalltime"] == 7
# the broadcast weights drive the library: one small descriptor pass
from yoho_amd import hip, synth
ctx = hip.Context(0)
ctx.load_partI(got)
x = torch.from_numpy(synth.unit_features(40, seed=3)).cuda()
o = ctx.partI_forward(x, want_inv=True)
torch.cuda.synchronize()
assert torch.isfinite(o["eqv"]).all()
ydist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def test_rccl_world1_collectives_take_the_device_branches():
    env = _env(RANK=0, WORLD_SIZE=1, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    p = subprocess.run([sys.executable, "-c", _WORKER], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORLD1_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_bench_under_the_launcher_with_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the launch line the driver uses for N > 1, with N = 1;
    the group is created (backend nccl), weights are broadcast, the timed regions are bracketed by RCCL barriers."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "1",
           "--no-cpu-baseline", "--no-dataset", "--no-yohoc", "--no-fcgf", "--no-sustained"]
    p = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    assert d["ranks"]["backend"] == "nccl" and d["ranks"]["process_group"] is True and len(d["ranks"]["ms_per_step_per_rank"]["all"]) == 1


# ---- more than one rank: ready to fire on any box with two devices -----------------------------------------------------------------
_WORKER2 = r"""
import hashlib, json, os, sys
import numpy as np, torch, torch.distributed as dist
from yoho_amd import dist as ydist, weights as W
rank, world, local = ydist.init_from_env("nccl")
assert world == 2 and local == rank and dist.get_backend() == "nccl" and torch.cuda.current_device() == rank
sd = W.synth_state_dict(W.PARTI_SPEC, 7)
# rank 1 starts from DIFFERENT weights: whatever it holds after the broadcast came over xGMI from rank 0's device
mine = sd if rank == 0 else W.synth_state_dict(W.PARTI_SPEC, 99)
got = ydist.broadcast_state_dict(mine, W.PARTI_SPEC)
for k, v in sd.items():                                      # tensor by tensor, on both ranks
    if not k.endswith("num_batches_tracked"):
        assert np.array_equal(got[k], np.asarray(v, np.float32)), (rank, k)
ydist.barrier()
assert ydist.all_ranks(10.0 + rank) == [10.0, 11.0] and ydist.max_over_ranks(float(rank)) == 1.0
g = ydist.gather_results({f"{rank}-9": {"trans": np.full((3, 4), float(rank)), "recalltime": rank}})
if rank == 0:
    assert len(g) == 2 and g[1]["1-9"]["recalltime"] == 1 and float(g[1]["1-9"]["trans"][0, 0]) == 1.0
# the sharded dataset driver over the two devices (tools/bench_dataset.run: scene cut over both ranks, one weight broadcast, host gather)
sys.path.insert(0, os.path.join(os.environ["YOHO_REPO"], "tools"))
import bench_dataset
out = bench_dataset.run(nfrag=10, kp=600, span=4, estimator="yohoo", workdir=os.environ["YOHO_WORKDIR"], runs=1, max_iter=100)
if rank == 0:
    print("RCCL_WORLD2 " + json.dumps({"sha": out.get("trans_sha256"), "pairs": out["pairs"], "rr": [r["registration_recall"] for r in out["runs"]]}))
ydist.barrier()
dist.destroy_process_group()
"""


def test_rccl_world2_broadcast_and_sharded_driver(tmp_path):
    """Two ranks on two devices over RCCL: the weight broadcast compared tensor by tensor on rank 1 (which starts from
    different weights), barrier / all-ranks / max / host gather across the ranks, and the sharded dataset driver over both devices
    = the world-1 digest (every pair's transform and recalltime).  A one-GPU box reports the skip with its reason; any box with two
    devices runs it - nothing else in the tree would put an ncclBroadcast between two GPUs before the driver's scaling run does."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip(f"skipped: {ndev} device - the world-2 RCCL test needs two (it runs unchanged on any multi-GPU box)")
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import bench_dataset
    one = bench_dataset.run(nfrag=10, kp=600, span=4, estimator="yohoo", workdir=str(tmp_path / "w1"), runs=1, max_iter=100)
    script = tmp_path / "worker2.py"
    script.write_text(_WORKER2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    p = subprocess.run(cmd, cwd=REPO, env=_env(YOHO_REPO=REPO, YOHO_WORKDIR=str(tmp_path / "w2")), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL_WORLD2 ")][-1]
    d = json.loads(line[len("RCCL_WORLD2 "):])
    assert d["pairs"] == one["pairs"] == 30 and d["sha"] == one["trans_sha256"], "world-2 transforms over RCCL differ from world-1"
    assert d["rr"] == [r["registration_recall"] for r in one["runs"]]


def test_bench_strong_scaling_line_reports_per_rank_work_and_plan():
    """`bench.py --scaling strong` (fixed total work cut over the ranks by the dataset driver's plan): the line carries every rank's
    pair count and time and the plan's predicted imbalance, so that a scaling run explains itself.  World 1 here; the same keys at N > 1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "1", "--scaling", "strong",
           "--no-cpu-baseline", "--no-dataset", "--no-yohoc", "--no-fcgf", "--no-sustained"]
    p = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1
    r = d["ranks"]
    assert len(r["pairs_per_rank"]) == 1 and len(r["ms_per_rank"]) == 1 and r["pairs_per_rank"][0] > 0 and r["ms_per_rank"][0] > 0
    assert r["plan_predicted_imbalance"] >= 1.0
