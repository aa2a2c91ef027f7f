"""PartII at M matches per arithmetic mode: event-timed passes + error against the f32 kernels: time_partII.py [M] [modes...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3233
modes = sys.argv[2:] or ["fp16x2", "cgemm", "cgemm8"]
ctx = hip.Context(0)
ctx.load_partII(W.synth_state_dict(W.PARTII_SPEC, 8))
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = [cu(synth.unit_features(M, seed=s)) for s in (1, 2, 3, 4)]
dr = cu(np.random.RandomState(0).randint(0, 60, size=M).astype(np.int64))
ctx.set_partII_mode("f32")
q32 = ctx.partII_forward(*args, dr)
for mode in modes:
    ctx.set_partII_mode(mode)
    for _ in range(3):
        q = ctx.partII_forward(*args, dr, check_range=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(3):
        e0.record()
        for _ in range(10):
            q = ctx.partII_forward(*args, dr, check_range=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print("%-8s M=%d  %.3f ms per pass (%s)  max |q - q_f32| %.3g  range %s" % (mode, M, min(ts), " ".join("%.3f" % t for t in ts), float((q - q32).abs().max()), ctx.range_status()), flush=True)
