"""Micro-benchmark of the FCGF backbone (csrc/sparse.hip): voxelise + forward on a synthetic surface cloud.
usage: bench_fcgf.py [points] [reps] [rotated copies per pass]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = hip.Context(0)
ctx.load_fcgf(W.synth_state_dict(W.FCGF_SPEC, 3))
pc = torch.from_numpy(synth.surface_cloud(n, seed=1, extent=3.0)).cuda()


def run():
    sel, coords = ctx.fcgf_voxelize(pc, 0.025)
    return coords, ctx.fcgf_forward(coords)


coords, F = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    sel, c = ctx.fcgf_voxelize(pc, 0.025)
torch.cuda.synchronize()
tv = (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
for _ in range(reps):
    F = ctx.fcgf_forward(coords)
torch.cuda.synchronize()
tf = (time.perf_counter() - t0) / reps * 1e3
nv = coords.shape[0]
# dense-equivalent work: 27-offset convs at every level (counted on the actual level sizes is not available here: level 0 only)
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if nb > 1:
    # the extractor's regime: nb rotated copies of the cloud in one backbone pass
    R = ctx.tables.R64
    group = [ctx.fcgf_voxelize_rotated(pc, R[i], 0.025)[1] for i in range(nb)]
    ctx.fcgf_forward_batch(group)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.fcgf_forward_batch(group)
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / reps * 1e3
    print(f"{nb} rotated copies, {sum(g.shape[0] for g in group)} voxels in one pass: {tb:.3f} ms ({tb / nb:.3f} ms per copy)")
    ctx.phase_profile(True)
    for _ in range(reps):
        ctx.fcgf_forward_batch(group)
    ph = ctx.phase_read()
    ctx.phase_profile(False)
    print("  phases, ms per pass: " + ", ".join(f"{k} {v['ms'] / reps:.3f}" for k, v in ph.items() if v["ms"] > 0))
import hashlib
print("sha256 of the single-cloud features:", hashlib.sha256(F.cpu().numpy().tobytes()).hexdigest()[:16])
print(f"points {n} -> voxels {nv}: voxelize {tv:.3f} ms, backbone forward {tf:.3f} ms  ({nv / tf * 1e-3:.2f} M voxels/s); "
      f"60 rotations of this cloud: {60 * (tv + tf):.1f} ms")
