"""Micro-benchmarks of the non-conv rows (gather, matcher, estimators) on realistic sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W
ctx = hip.Context(0)
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
rs = np.random.RandomState(0)
K = 5000
keys = torch.from_numpy(rs.rand(K,3)*3).cuda()
out = torch.zeros((K,32,60), device="cuda")
for n in (10000, 50000, 100000):
    pts = torch.from_numpy((rs.rand(n,3)*3).astype(np.float32)).cuda()
    feat = torch.from_numpy(rs.randn(n,32).astype(np.float32)).cuda()
    ms = timeit(lambda: ctx.group_gather(keys, pts, feat, 7, out))
    print(f"group_gather K=5000 n={n}: {ms:.3f} ms per group element -> {60*ms:.1f} ms per fragment")
    q = torch.from_numpy((rs.rand(K,3)*3).astype(np.float32)).cuda()
    ms = timeit(lambda: ctx.nn_search(q, pts, want_dist=False, squared=True))
    print(f"nn_search D=3 (fp32 SquareL2) K=5000 n={n}: {ms:.3f} ms")
a = torch.from_numpy(rs.randn(K,32).astype(np.float32)).cuda(); b = torch.from_numpy(rs.randn(K,32).astype(np.float32)).cuda()
print("mutual_nn 5000x5000x32: %.3f ms" % timeit(lambda: ctx.mutual_nn(a,b)))
ec = synth.estimator_case(1500, 1000, seed=4)
cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
k0,k1,T = cu(ec["k0"]),cu(ec["k1"]),cu(ec["T"])
order = cu(np.arange(1500))
print("yoho_o_score M=1500 H=1000: %.3f ms" % timeit(lambda: ctx.o_score(k0,k1,T,order,1000,0.09)))
tri = cu(rs.randint(0,1500,(1000,3)).astype(np.int64))
print("yoho_c_ransac M=1500 I=1000: %.3f ms" % timeit(lambda: ctx.c_ransac(k0,k1,tri,None,0.07)))
d1 = torch.from_numpy(synth.unit_features(1500, seed=1)).cuda(); d2 = torch.from_numpy(synth.unit_features(1500, seed=2)).cuda()
print("des2r M=1500: %.3f ms" % timeit(lambda: ctx.des2r(d1,d2)))
