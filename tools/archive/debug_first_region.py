"""Why is the first timed region of bench.py slower than the next ones?  Times chunks of 5 steps back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W, pipeline
dev = 0
torch.cuda.set_device(dev)
sd1 = W.synth_state_dict(W.PARTI_SPEC, 7); sd2 = W.synth_state_dict(W.PARTII_SPEC, 8)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
pr = synth.make_pair(5000, seed=10)
pair = (cu(pr["feat0"]), cu(pr["feat1"]), cu(pr["keys0"]), cu(pr["keys1"]))
st = pipeline.PairStreamer(lambda: hip.Context(dev), sd1, sd2)
rng = np.random.RandomState(1234)
def run(n, seed0):
    return st.run([pair] * n, inlier_dist=0.09, max_iter=1000, order_rng=rng, estimator="yohoo", seeds=[seed0 + i for i in range(n)])[-1]
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
run(5, 1)
torch.cuda.synchronize()
print("mem allocated / reserved GB after warm-up: %.2f / %.2f" % (torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "chunks":
        ts = []
        for c in range(4):
            run(5, 1000 + 100 * rep + 5 * c)
            torch.cuda.synchronize()
            ts.append(time.perf_counter())
        print("rep", rep, "5-step chunks ms/step:", [round((b - a) / 5 * 1e3, 3) for a, b in zip([t0] + ts[:-1], ts)])
    else:
        run(20, 1000 + 100 * rep)
        torch.cuda.synchronize()
        print("rep", rep, "ms/step %.3f" % ((time.perf_counter() - t0) / 20 * 1e3), "reserved GB %.2f" % (torch.cuda.memory_reserved() / 1e9))
