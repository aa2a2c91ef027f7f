#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fcgf.py tests/test_gpu_dropin.py -q -x -k "fcgf or extractor or testset or backbone or duplicate or row_orders or batched" > $O/pytest_gpu_g.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu_g.log
for m in staged fused; do echo "== YOHO_FCGF_NORM=$m"; YOHO_FCGF_NORM=$m timeout 300 python tools/bench_extract.py 300000 5000 2>&1 | tail -2; done
python - <<'PY'
import os, sys, subprocess
code = r'''
import sys, hashlib
sys.path.insert(0, ".")
import numpy as np, torch
from yoho_amd import hip, synth, weights as W
ctx = hip.Context(0)
ctx.load_fcgf(W.synth_state_dict(W.FCGF_SPEC, 3))
pc = torch.from_numpy(synth.surface_cloud(120000, seed=5, extent=2.5)).cuda()
sel, coords = ctx.fcgf_voxelize(pc, 0.025)
F = ctx.fcgf_forward(coords)
small = coords[:9000].contiguous()
Fs = ctx.fcgf_forward(small)                      # < 32768 voxels: the stand-alone normalise kernel
Fb = ctx.fcgf_forward_batch([small, coords])      # the same cloud inside a large pass: normalised in the epilogue
h = hashlib.sha256(); h.update(F.cpu().numpy().tobytes()); h.update(Fs.cpu().numpy().tobytes())
print(h.hexdigest()[:16], bool(torch.equal(Fs, Fb[0])), bool(torch.equal(F, Fb[1])), float((F.norm(dim=1) - 1).abs().max()))
'''
for m in ("staged", "fused"):
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, YOHO_FCGF_NORM=m), capture_output=True, text=True)
    print(m, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-600:])
PY
