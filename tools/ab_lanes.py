"""A/B of yoho_extractor's backbone lanes (YOHO_FCGF_LANES): wall ms per fragment with one and two lanes, alternating, and
bit equality of what the two return.  usage: ab_lanes.py [points] [keypoints] [rounds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import synth, weights as W
from yoho_amd.yoho_extract import yoho_extractor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7}, "state_dict": fsd}
ex = yoho_extractor(fcgf_ckpt=ck, yoho_ckpt=W.synth_state_dict(W.PARTI_SPEC, 7))
pc = synth.surface_cloud(n, seed=1, extent=3.0)
out = {}
times = {1: [], 2: []}
for rep in range(2 * rounds + 2):
    lanes = 1 + rep % 2
    ex.lanes = lanes
    np.random.seed(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    kpts, inv, eqv = ex.run(pc, voxel_size=0.025, nkpts=nk)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    if rep >= 2:
        times[lanes].append(dt)
    out[lanes] = (kpts.copy(), inv.clone(), eqv.clone(), ex._last_group_feats.clone())
    print(f"run {rep}: lanes {lanes}: {dt:.2f} ms", flush=True)
for l in (1, 2):
    t = sorted(times[l])
    print(f"lanes {l}: median {t[len(t) // 2]:.2f} ms, min {t[0]:.2f}, max {t[-1]:.2f}")
same = all(bool((a == b).all()) if isinstance(a, np.ndarray) else torch.equal(a, b) for a, b in zip(out[1], out[2]))
print("outputs of the two modes bit-identical:", same)
