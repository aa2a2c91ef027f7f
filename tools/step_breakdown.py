"""Wall-clock breakdown of one bench step (pipeline.run_pair) into its stages, with a device sync after each stage."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W

KP = 5000
ctx = hip.Context(0)
ctx.load_partI(W.synth_state_dict(W.PARTI_SPEC, 7))
ctx.load_partII(W.synth_state_dict(W.PARTII_SPEC, 8))
pr = synth.make_pair(KP, seed=10)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
f0, f1, k0, k1 = cu(pr["feat0"]), cu(pr["feat1"]), cu(pr["keys0"]), cu(pr["keys1"])
rng = np.random.RandomState(0)
acc = {}


def lap(name, t0):
    torch.cuda.synchronize()
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t


N = 10
for it in range(N + 2):
    if it == 2:
        acc.clear()
    torch.cuda.synchronize()
    t = time.perf_counter()
    o = ctx.partI_forward(torch.cat([f0, f1]), want_inv=False, want_inv_np=True)
    o0 = {k: v[:KP] for k, v in o.items()}
    o1 = {k: v[KP:] for k, v in o.items()}
    t = lap("partI (both fragments)", t)
    match = ctx.mutual_nn(o0["inv_np"], o1["inv_np"])
    t = lap("mutual_nn", t)
    M = match.shape[0]
    m0, m1 = match[:, 0], match[:, 1]
    k0m, k1m = k0[m0].contiguous(), k1[m1].contiguous()
    t = lap("key gathers", t)
    dr = ctx.des2r_matched(o1["eqv"], o0["eqv"], match)
    t = lap("des2r", t)
    q = ctx.partII_forward_matched(f0, f1, o0["eqv"], o1["eqv"], match, dr)
    t = lap("partII", t)
    T = ctx.hyp_from_quat(q, dr, k0m, k1m)
    t = lap("hyp_from_quat", t)
    order = np.arange(M)
    rng.shuffle(order)
    od = torch.from_numpy(order).to(f0.device)
    t = lap("order shuffle+upload", t)
    res, _ = ctx.o_score(k0m, k1m, T, od, min(1000, M), 0.09)
    t = lap("o_score", t)
    bh, bc = (int(v) for v in res.cpu().numpy())
    Tb = T[int(order[bh])].cpu().numpy()
    t = lap("readback", t)
print("matches:", M)
tot = 0.0
for k, v in acc.items():
    print("%-22s %8.3f ms" % (k, v / N * 1e3))
    tot += v / N * 1e3
print("%-22s %8.3f ms (with a sync after every stage)" % ("total", tot))
