"""PartI pass timing (event-timed per-launch breakdown) for one gconv mode: time_partI.py [mode] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W
mode = sys.argv[1] if len(sys.argv) > 1 else "fgemm"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ctx = hip.Context(0)
ctx.load_partI(W.synth_state_dict(W.PARTI_SPEC, 7))
ctx.set_gconv_mode(mode)
x = torch.from_numpy(synth.unit_features(B, seed=1)).cuda()
for _ in range(3):
    ctx.partI_forward(x, want_inv=False, want_inv_np=True)
ctx.set_profiling(True)
ms = []
for _ in range(5):
    ctx.partI_forward(x, want_inv=False, want_inv_np=True)
    torch.cuda.synchronize()
    ms.append([ctx.kernel_ms(i) for i in range(7)])
ms = np.array(ms).mean(0)
print(mode, "B=%d" % B, "conv launches", np.round(ms[:4], 3), "head %.3f tail %.3f transforms %.3f total %.3f" % (ms[4], ms[5], ms[6], ms.sum()),
      )
