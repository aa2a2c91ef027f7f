#!/bin/bash
# FCGF path check on the GPU box: the fcgf / extractor tests, then - in separate processes, the switches are read once - the two
# coordinate-map paths (rank-ordered bitmaps / hash tables) on voxelisation and backbone outputs (sha256 must agree), and the extractor's time
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_fcgf.py tests/test_gpu_dropin.py -q -x -k "fcgf or extractor or testset or backbone or voxel or duplicate or row_orders" 2>&1 | tail -3
python - <<'PY'
import os, sys, subprocess
code = r'''
import sys, hashlib
sys.path.insert(0, ".")
import numpy as np, torch
from yoho_amd import hip, synth, weights as W
ctx = hip.Context(0)
ctx.load_fcgf(W.synth_state_dict(W.FCGF_SPEC, 3))
pc = torch.from_numpy(synth.surface_cloud(200000, seed=5, extent=2.5) - 1.3).cuda()
R = ctx.tables.R64
h = hashlib.sha256()
outs = ctx.fcgf_voxelize_rotated_batch(pc, [R[i] for i in range(15)], 0.025)
for sel, coords, ps in outs:
    for t in (sel, coords, ps): h.update(t.cpu().numpy().tobytes())
for off, k in ((40.0, 3), (0.0, 1)):
    for sel, coords, ps in ctx.fcgf_voxelize_rotated_batch(pc[:5000] + off, [R[i] for i in range(k)], 0.025):
        for t in (sel, coords, ps): h.update(t.cpu().numpy().tobytes())
F = ctx.fcgf_forward_batch([o[1] for o in outs])
for f in F: h.update(f.cpu().numpy().tobytes())
print(h.hexdigest()[:16], [int(o[0].shape[0]) for o in outs][:4])
'''
res = {}
for m in ("hash", "rank"):
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, YOHO_FCGF_COORDS=m), capture_output=True, text=True)
    res[m] = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-800:]
    print(m, res[m])
print("IDENTICAL" if res["hash"] == res["rank"] else "DIFFERENT")
PY
for m in hash rank; do echo "== YOHO_FCGF_COORDS=$m"; YOHO_FCGF_COORDS=$m timeout 300 python tools/bench_extract.py 300000 5000 2>&1 | tail -2; done
