"""Workload for the PMC passes: a few PartI forward passes on 5000 keypoints (bf16x3 and fp32 modes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W
ctx = hip.Context(0)
ctx.load_partI(W.synth_state_dict(W.PARTI_SPEC, 7))
B = int(os.environ.get("PMC_B", "5000"))
x = torch.from_numpy(synth.unit_features(B, seed=1)).cuda()
for mode in (sys.argv[1:] or ["fgemm", "fourier", "bf16x3", "f32"]):
    ctx.set_gconv_mode(mode)
    for _ in range(2):
        ctx.partI_forward(x, want_inv=False, want_inv_np=True)
torch.cuda.synchronize()
