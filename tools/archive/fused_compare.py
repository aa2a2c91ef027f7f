"""fgemm3's K loops alone (coefficient stores switched off: experiments build) beside the normal launches, under sustained load -
the like-for-like partner of tools/fused_tile_probe.hip, which has no stores either.
    YOHO_LIB=exp python tools/fused_compare.py        (needs YOHO_EXPERIMENTS=1 python -m yoho_amd.build)"""
import os, sys
os.environ.setdefault("YOHO_LIB", "exp")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = hip.Context(0)
ctx.load_partI(W.synth_state_dict(W.PARTI_SPEC, 7))
x = torch.from_numpy(synth.unit_features(B, seed=1)).cuda()
ctx.set_profiling(True)
for label, env in (("with coefficient stores (as shipped)", ""), ("K loops alone (YOHO_FGEMM_DEBUG=nostore)", "nostore"), ("with coefficient stores (as shipped)", "")):
    os.environ["YOHO_FGEMM_DEBUG"] = env
    rows = []
    for _ in range(3):
        for _ in range(9):                       # eight passes back to back, the ninth is read: the clock of sustained load
            ctx.partI_forward(x, want_inv=False, want_inv_np=True, check_range=False)
        torch.cuda.synchronize()
        rows.append([ctx.kernel_ms(i) for i in range(13)])
    ms = np.array(rows).mean(0)
    print("%-46s B=%d  fgemm launches 32->256 %.3f  256->512 %.3f  512->256 %.3f  256->32 %.3f   transforms %.3f %.3f %.3f   pass %.3f ms"
          % (label, B, ms[0], ms[1], ms[2], ms[3], ms[7], ms[8], ms[9], ms[12]))
