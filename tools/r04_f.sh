#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fcgf.py tests/test_gpu_dropin.py -q -x -k "fcgf or extractor or testset or backbone" > $O/pytest_gpu_f.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu_f.log
for m in hash rank; do echo "== YOHO_FCGF_COORDS=$m"; YOHO_FCGF_COORDS=$m timeout 300 python tools/bench_extract.py 300000 5000 2>&1 | tail -2; YOHO_FCGF_COORDS=$m timeout 120 python tools/bench_fcgf.py 300000 5 15 2>&1 | tail -2; done
python - <<'PY'
# bit-identity of the two coordinate-map paths on the backbone output (one cloud and a 15-copy pass), plus a cloud with duplicate voxels
import os, sys, subprocess, hashlib
code = r'''
import sys, os, hashlib
sys.path.insert(0, ".")
import numpy as np, torch
from yoho_amd import hip, synth, weights as W
ctx = hip.Context(0)
ctx.load_fcgf(W.synth_state_dict(W.FCGF_SPEC, 3))
pc = torch.from_numpy(synth.surface_cloud(120000, seed=5, extent=2.5)).cuda()
sel, coords = ctx.fcgf_voxelize(pc, 0.025)
F = ctx.fcgf_forward(coords)
R = ctx.tables.R64
group = [ctx.fcgf_voxelize_rotated(pc, R[i], 0.025)[1] for i in range(15)]
Fb = ctx.fcgf_forward_batch(group)
dup = torch.cat([coords[:1000], coords])            # duplicate voxels: the hash path keeps the first of each
try:
    Fd = ctx.fcgf_forward(dup)
    dd = hashlib.sha256(Fd.cpu().numpy().tobytes()).hexdigest()[:16]
except Exception as e:
    dd = "error: " + str(e)[:80]
h = hashlib.sha256()
h.update(F.cpu().numpy().tobytes())
for f in Fb: h.update(f.cpu().numpy().tobytes())
print(h.hexdigest()[:16], coords.shape[0], sum(g.shape[0] for g in group), dd)
'''
out = {}
for m in ("hash", "rank"):
    env = dict(os.environ, YOHO_FCGF_COORDS=m)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    out[m] = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-800:]
    print(m, out[m])
print("IDENTICAL" if out["hash"] == out["rank"] else "DIFFERENT")
PY
