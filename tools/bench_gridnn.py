"""Brute force vs hash-grid 1-NN on the feature-transfer workload: 5000 keypoints of a 300k-point surface cloud against its
voxel-downsampled copy (one point per 2.5 cm voxel), f64 group gather (YOHO_testset) and fp32 SquareL2 (yoho_extractor)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth
ctx = hip.Context(0)
def timeit(f, n=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
cloud = synth.surface_cloud(300000, seed=1, extent=3.0)
_, first = np.unique(np.floor(cloud / 0.025).astype(np.int64), axis=0, return_index=True)
pts = torch.from_numpy(cloud[np.sort(first)].astype(np.float32)).cuda()
rs = np.random.RandomState(0)
kidx = rs.permutation(len(cloud))[:5000]
keys = torch.from_numpy(cloud[kidx]).cuda()
q = keys.float().contiguous()
feat = torch.from_numpy(rs.randn(pts.shape[0], 32).astype(np.float32)).cuda()
out = torch.zeros((5000, 32, 60), device="cuda")
for cell in (0.0, 0.025):
    ctx.set_nn_grid(cell)
    a = timeit(lambda: ctx.group_gather(keys, pts, feat, 0, out))
    b = timeit(lambda: ctx.nn_search(q, pts, want_dist=False, squared=True))
    print(f"{'grid %.3f' % cell if cell else 'brute force'}: n={pts.shape[0]} group_gather (f64) {a:.3f} ms, nn_search (f32 SquareL2) {b:.3f} ms per group element")
