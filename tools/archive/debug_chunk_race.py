"""Which outputs of the two-stream depth-first PartI pass differ from the breadth-first pass, and under which conditions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import hip, synth, weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = hip.Context(0)
ctx.load_partI(W.synth_state_dict(W.PARTI_SPEC, 7))
x = torch.from_numpy(synth.unit_features(B, seed=1)).cuda()
ctx.set_partI_schedule(0, 1)
ref = ctx.partI_forward(x, want_inv=False, want_inv_np=True)["eqv"].clone()
torch.cuda.synchronize()
REPS = int(os.environ.get('RACE_REPS', '6'))
SCH = [tuple(int(v) for v in t.split('x')) for t in os.environ.get('RACE_SCHEDS', '1024x1,1024x2,4096x2,2048x2,512x2,2560x2,5120x2').split(',')]
nbad = 0
for chunk, ns in SCH:
    ctx.set_partI_schedule(chunk, ns)
    for rep in range(REPS):
        o = ctx.partI_forward(x, want_inv=False, want_inv_np=True)["eqv"]
        torch.cuda.synchronize()
        bad = (o != ref).reshape(B, -1).any(1).nonzero().flatten().cpu().numpy()
        if len(bad):
            nbad += 1
            ch = np.unique(bad // chunk)
            d = (o - ref).abs()
            print(f"chunk {chunk} x{ns} rep {rep}: {len(bad)} rows differ, chunks {ch.tolist()}, rows {bad[:6].tolist()}..{bad[-3:].tolist()}, "
                  f"max abs {float(d.max()):.3g}, nan {int(torch.isnan(o).sum())}, elems {int((o != ref).sum())}", flush=True)
        else:
            print(f"chunk {chunk} x{ns} rep {rep}: identical", flush=True)
print('passes with differences:', nbad, 'of', REPS * len(SCH))
