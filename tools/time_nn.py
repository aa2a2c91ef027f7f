"""mutual NN timing: pre-filter vs brute force on the bench's descriptors (5000 x 5000)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth
c = hip.Context(0)
pr = synth.make_pair(5000, seed=10)
a = torch.from_numpy(np.ascontiguousarray(np.mean(pr["feat0"], -1))).cuda()
b = torch.from_numpy(np.ascontiguousarray(np.mean(pr["feat1"], -1))).cuda()
for on in (True, False, True, False):
    c.set_nn_prefilter(on)
    for _ in range(3):
        m = c.mutual_nn(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m = c.mutual_nn(a, b)
    e1.record()
    torch.cuda.synchronize()
    print("prefilter" if on else "brute    ", "%.3f ms per call, %d matches" % (e0.elapsed_time(e1) / 20, m.shape[0]))
