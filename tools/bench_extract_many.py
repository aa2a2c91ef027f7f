"""yoho_extractor.run in a loop against yoho_extractor.run_many (streamed): ms per fragment and equality of everything returned.
usage: bench_extract_many.py [fragments] [points] [keypoints]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import synth, weights as W
from yoho_amd.yoho_extract import yoho_extractor

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
nk = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7}, "state_dict": fsd}
ex = yoho_extractor(fcgf_ckpt=ck, yoho_ckpt=W.synth_state_dict(W.PARTI_SPEC, 7))
clouds = [synth.surface_cloud(n, seed=1 + i, extent=3.0) for i in range(3)]
pcs = [clouds[i % 3] for i in range(nf)]
res = {}
for rep in range(4):
    mode = "run" if rep % 2 == 0 else "run_many"
    np.random.seed(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = [ex.run(pc, voxel_size=0.025, nkpts=nk) for pc in pcs] if mode == "run" else list(ex.run_many(pcs, voxel_size=0.025, nkpts=nk))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[mode] = out
    print(f"rep {rep}: {mode}: {nf} fragments in {dt * 1e3:.1f} ms = {dt / nf * 1e3:.2f} ms per fragment", flush=True)
same = all(np.array_equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) for a, b in zip(res["run"], res["run_many"]))
print("run_many returns what run returns, fragment by fragment:", same)
