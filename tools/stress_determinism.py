"""Stress check of the kernels whose loops keep memory operations in flight across barriers with counted waits (gft16x, fgemm3 /
fgemm3s, cone1, gconv16): a race would show up as a sporadic bit difference.  PartI: the default mode against fgemm256 (other
GEMM / transform kernels, same arithmetic) on fresh random inputs; PartII and the FCGF backbone: run-to-run determinism.
usage: stress_determinism.py [reps=30]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = hip.Context(0)
ctx.load_partI(W.synth_state_dict(W.PARTI_SPEC, 7))
ctx.load_partII(W.synth_state_dict(W.PARTII_SPEC, 8))
bad = 0
for rep in range(reps):
    B = [10000, 5000, 4097, 777][rep % 4]
    x = torch.from_numpy(synth.unit_features(B, seed=100 + rep)).cuda()
    ctx.set_gconv_mode("fgemm")
    a = ctx.partI_forward(x, want_inv=True)
    a2 = ctx.partI_forward(x, want_inv=True)
    ctx.set_gconv_mode("fgemm256")
    b = ctx.partI_forward(x, want_inv=True)
    ok = torch.equal(a["eqv"], b["eqv"]) and torch.equal(a["eqv"], a2["eqv"]) and torch.equal(a["inv"], b["inv"])
    bad += 0 if ok else 1
    if not ok:
        print("PartI mismatch at rep", rep, "B", B, (a["eqv"] - b["eqv"]).abs().max().item())
ctx.set_gconv_mode("fgemm")
print("PartI: %d reps, %d mismatches" % (reps, bad))
# the opt-in fp8-correction arithmetic (fgemm3c: counted waits across barriers too, plus a device-side scale word): run-to-run determinism
bad8 = 0
ctx.set_gconv_mode("fgemm8")
for rep in range(reps):
    B = [10000, 5000, 4097, 777][rep % 4]
    x = torch.from_numpy(synth.unit_features(B, seed=100 + rep)).cuda()
    a = ctx.partI_forward(x, want_inv=True)
    a2 = ctx.partI_forward(x, want_inv=True)
    ok = torch.equal(a["eqv"], a2["eqv"]) and torch.equal(a["inv"], a2["inv"]) and bool(torch.isfinite(a["eqv"]).all())
    bad8 += 0 if ok else 1
    if not ok:
        print("PartI fgemm8 nondeterminism at rep", rep, "B", B, (a["eqv"] - a2["eqv"]).abs().max().item())
ctx.set_gconv_mode("fgemm")
print("PartI fgemm8: %d reps, %d mismatches" % (reps, bad8))

# PartII determinism on a pair's matches
pr = synth.make_pair(5000, seed=3)
f0, f1 = torch.from_numpy(pr["feat0"]).cuda(), torch.from_numpy(pr["feat1"]).cuda()
e0 = ctx.partI_forward(f0, want_inv=False)["eqv"]; e1 = ctx.partI_forward(f1, want_inv=False)["eqv"]
M = 3200
g = torch.Generator(device="cpu").manual_seed(1)
i0 = torch.randint(0, 5000, (M,), generator=g).cuda(); i1 = torch.randint(0, 5000, (M,), generator=g).cuda()
dr = torch.randint(0, 60, (M,), generator=g).cuda()
ref = None; bad2 = 0
for rep in range(reps):
    q = ctx.partII_forward(f1[i1].contiguous(), f0[i0].contiguous(), e1[i1].contiguous(), e0[i0].contiguous(), dr)
    if ref is None: ref = q.clone()
    elif not torch.equal(q, ref):
        bad2 += 1; print("PartII nondeterminism at rep", rep, (q - ref).abs().max().item())
print("PartII: %d reps, %d mismatches" % (reps, bad2))

# FCGF backbone determinism (batched pass)
ctx.load_fcgf(W.synth_state_dict(W.FCGF_SPEC, 3))
pc = torch.from_numpy(synth.surface_cloud(120000, seed=2, extent=2.5)).cuda()
R = ctx.tables.R64
group = [ctx.fcgf_voxelize_rotated(pc, R[i], 0.025)[1] for i in range(6)]
ref = None; bad3 = 0
for rep in range(max(4, reps // 4)):
    out = torch.cat(ctx.fcgf_forward_batch(group))
    if ref is None: ref = out.clone()
    elif not torch.equal(out, ref):
        bad3 += 1; print("FCGF nondeterminism at rep", rep, (out - ref).abs().max().item())
print("FCGF: %d mismatches" % bad3)
# two pairs in flight at full size (what bench.py times): streamed results against the sequential pipeline, twice
from yoho_amd import pipeline
sd1, sd2 = W.synth_state_dict(W.PARTI_SPEC, 7), W.synth_state_dict(W.PARTII_SPEC, 8)
cu = lambda a: torch.from_numpy(a).cuda()
prs = [synth.make_pair(5000, seed=60 + i) for i in range(3)]
pairs = [(cu(p["feat0"]), cu(p["feat1"]), cu(p["keys0"]), cu(p["keys1"])) for p in prs] * 3
st = pipeline.PairStreamer(lambda: hip.Context(0), sd1, sd2)
bad4 = 0
runs = [st.run(pairs, inlier_dist=0.09, max_iter=1000, order_rng=np.random.RandomState(5)) for _ in range(2)]
rng = np.random.RandomState(5)
for i, p in enumerate(pairs):
    ref = pipeline.run_pair(ctx, *p, inlier_dist=0.09, max_iter=1000, order_rng=rng)
    for got in (runs[0][i], runs[1][i]):
        if not (torch.equal(got.match, ref.match) and torch.equal(got.dr_index, ref.dr_index) and torch.equal(got.quat, ref.quat)
                and np.array_equal(got.trans, ref.trans) and got.best_count == ref.best_count):
            bad4 += 1; print("streamed pair", i, "differs from the sequential pipeline")
print("PairStreamer at 5000 keypoints: %d pairs x 2 runs, %d mismatches" % (len(pairs), bad4))
sys.exit(1 if bad or bad2 or bad3 or bad4 else 0)
