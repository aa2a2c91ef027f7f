"""End-to-end one-cloud API (yoho_extractor.run): cloud -> 60 x (rotate, voxelise, FCGF backbone, NN transfer) -> PartI.
usage: bench_extract.py [points] [keypoints]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import synth, weights as W
from yoho_amd.yoho_extract import yoho_extractor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7}, "state_dict": fsd}
ex = yoho_extractor(fcgf_ckpt=ck, yoho_ckpt=W.synth_state_dict(W.PARTI_SPEC, 7))
pc = synth.surface_cloud(n, seed=1, extent=3.0)
for rep in range(3):
    np.random.seed(rep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    kpts, inv, eqv = ex.run(pc, voxel_size=0.025, nkpts=nk)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {rep}: {n} points, {nk} keypoints -> eqv {tuple(eqv.shape)} in {dt * 1e3:.1f} ms")
