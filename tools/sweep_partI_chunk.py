"""Depth-first PartI schedule sweep: one 10000-keypoint pass of the default mode breadth-first and cut into
chunks of 512 ... 4096 keypoints on one stream or alternating over two (yoho_set_partI_schedule).

    python tools/sweep_partI_chunk.py [B=10000] [out.json]

Per schedule: whole-pass time (HIP events around 10 back-to-back passes), the per-launch sums of a profiled pass (GEMMs,
transforms), the shader clock the part holds meanwhile (library clock probe on a high-priority stream + SMU samples), and whether
the outputs are bit-identical to the breadth-first pass.  PMC traffic per schedule comes from tools/collect_chunk_sweep.sh.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yoho_amd import hip, synth, weights as W
from yoho_amd.power import PowerMonitor, ClockProbe

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
ctx = hip.Context(0)
ctx.load_partI(W.synth_state_dict(W.PARTI_SPEC, 7))
x = torch.from_numpy(synth.unit_features(B, seed=1)).cuda()
NPASS = 12
scheds = [(0, 1)] + [(c, s) for s in (1, 2) for c in (512, 1024, 2048, 4096) if c < B]
if os.environ.get("SWEEP_SCHEDS"):
    scheds = [tuple(int(v) for v in t.split("x")) for t in os.environ["SWEEP_SCHEDS"].split(",")]
ref = None
rows = []
mon = PowerMonitor(0)
for chunk, nstr in scheds:
    ctx.set_partI_schedule(chunk, nstr)
    for _ in range(3):
        o = ctx.partI_forward(x, want_inv=False, want_inv_np=True)
    torch.cuda.synchronize()
    same = None
    if ref is None:
        ref = {k: v.clone() for k, v in o.items()}
    else:
        same = all(torch.equal(o[k], ref[k]) for k in ref)
    # whole-pass time and the clock under this load
    probe = ClockProbe(ctx, us=500)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    mon.start()
    e0.record()
    probe.queue(2)
    for i in range(NPASS):
        ctx.partI_forward(x, want_inv=False, want_inv_np=True, check_range=False)
        probe.queue(8)
    e1.record()
    torch.cuda.synchronize()
    smu = mon.stop()
    pass_ms = e0.elapsed_time(e1) / NPASS
    clk = probe.mhz()
    # per-launch sums (a profiled pass; with two streams the launches of neighbouring chunks overlap, so their sum exceeds the pass)
    ctx.set_profiling(True)
    ms = []
    for _ in range(3):
        ctx.partI_forward(x, want_inv=False, want_inv_np=True, check_range=False)
        torch.cuda.synchronize()
        ms.append([ctx.kernel_ms(i) for i in range(13)])
    ctx.set_profiling(False)
    ms = np.array(ms).mean(0)
    row = {"chunk_kp": chunk, "streams": nstr, "pass_ms": round(pass_ms, 4), "profiled_pass_ms": round(float(ms[12]), 4),
           "gemm_ms": [round(float(v), 4) for v in ms[:4]], "gemm_sum_ms": round(float(ms[:4].sum()), 4),
           "transform_ms": round(float(ms[6]), 4), "head_ms": round(float(ms[4]), 4), "tail_ms": round(float(ms[5]), 4),
           "probe_mhz_mean": round(float(np.mean(clk)), 1) if clk else None, "probe_mhz_min": round(float(np.min(clk)), 1) if clk else None,
           "probe_mhz_max": round(float(np.max(clk)), 1) if clk else None, "probes": len(clk),
           "smu": smu, "bit_identical_to_breadth_first": same}
    rows.append(row)
    print(json.dumps(row), flush=True)

print("\n| chunk kp | streams | pass ms | GEMM sum ms | transforms ms | probe MHz (mean / min) | SMU sclk MHz | SMU W | bits |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %d | %.3f | %.3f | %.3f | %s / %s | %s | %s | %s |" % (
        r["chunk_kp"] or "all", r["streams"], r["pass_ms"], r["gemm_sum_ms"], r["transform_ms"], r["probe_mhz_mean"], r["probe_mhz_min"],
        r["smu"]["sclk_mhz_mean"], r["smu"]["power_w_mean"], {None: "ref", True: "same", False: "DIFFER"}[r["bit_identical_to_breadth_first"]]))
if out_path:
    json.dump({"B": B, "passes_timed": NPASS, "rows": rows}, open(out_path, "w"), indent=1)
