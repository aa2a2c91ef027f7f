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
NPASS = 40
WARM_S = float(os.environ.get("SWEEP_WARM_S", "1.0"))
scheds = [(c, s) for s in (1, 2) for c in (512, 1024, 2048, 4096) if c < B] + [(c, 2) for c in (3072, 5120) if c < B]
if os.environ.get("SWEEP_SCHEDS"):
    scheds = [tuple(int(v) for v in t.split("x")) for t in os.environ["SWEEP_SCHEDS"].split(",")]
mon = PowerMonitor(0)
probe = ClockProbe(ctx, us=1000)
ctx.set_partI_schedule(0, 1)
ref = {k: v.clone() for k, v in ctx.partI_forward(x, want_inv=False, want_inv_np=True).items()}
torch.cuda.synchronize()


def measure(chunk, nstr, warm_s):
    """sustained load first (clock and power settle: the SMU's power reading is a moving average over ~1 s and the clock follows
    it), then NPASS back-to-back passes between two events, with SMU samples and 1 ms clock probes on their own stream meanwhile;
    then three profiled passes for the per-launch sums"""
    ctx.set_partI_schedule(chunk, nstr)
    t0 = time.perf_counter()
    o = None
    while time.perf_counter() - t0 < warm_s:
        for _ in range(10):
            o = ctx.partI_forward(x, want_inv=False, want_inv_np=True, check_range=False)
        torch.cuda.synchronize()
    same = all(torch.equal(o[k], ref[k]) for k in ref)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    mon.start()
    probe.queue(int(NPASS * 5.5))
    e0.record()
    for i in range(NPASS):
        ctx.partI_forward(x, want_inv=False, want_inv_np=True, check_range=False)
    e1.record()
    torch.cuda.synchronize()
    smu = mon.stop()
    pass_ms = e0.elapsed_time(e1) / NPASS
    clk = probe.mhz()
    ctx.set_profiling(True)
    ms = []
    for _ in range(3):
        ctx.partI_forward(x, want_inv=False, want_inv_np=True, check_range=False)
        torch.cuda.synchronize()
        ms.append([ctx.kernel_ms(i) for i in range(13)])
    ctx.set_profiling(False)
    ms = np.array(ms).mean(0)
    return {"chunk_kp": chunk, "streams": nstr, "pass_ms": round(pass_ms, 4), "profiled_pass_ms": round(float(ms[12]), 4),
            "gemm_ms": [round(float(v), 4) for v in ms[:4]], "gemm_sum_ms": round(float(ms[:4].sum()), 4),
            "transform_ms": round(float(ms[6]), 4), "head_ms": round(float(ms[4]), 4), "tail_ms": round(float(ms[5]), 4),
            "probe_mhz_mean": round(float(np.mean(clk)), 1) if clk else None, "probe_mhz_min": round(float(np.min(clk)), 1) if clk else None,
            "probe_mhz_max": round(float(np.max(clk)), 1) if clk else None, "probes": len(clk),
            "smu": smu, "bit_identical_to_breadth_first": same}


rows = []
measure(0, 1, 2 * WARM_S)                                   # heat-up, discarded
for chunk, nstr in scheds:
    base = measure(0, 1, 0.5 * WARM_S)                      # the breadth-first pass right before every schedule (drift control)
    row = measure(chunk, nstr, WARM_S)
    row["breadth_first_before"] = {k: base[k] for k in ("pass_ms", "gemm_sum_ms", "transform_ms", "probe_mhz_mean", "smu")}
    row["pass_ratio_to_breadth_first"] = round(row["pass_ms"] / base["pass_ms"], 4)
    rows.append(row)
    print(json.dumps(row), flush=True)

print("\n| chunk kp | streams | pass ms | breadth-first before: pass ms | ratio | GEMM sum ms (bf) | transforms ms (bf) | probe MHz (bf) | SMU sclk MHz / W (bf) | bits |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    b = r["breadth_first_before"]
    print("| %d | %d | %.3f | %.3f | %.3f | %.3f (%.3f) | %.3f (%.3f) | %s (%s) | %s / %s (%s / %s) | %s |" % (
        r["chunk_kp"], r["streams"], r["pass_ms"], b["pass_ms"], r["pass_ratio_to_breadth_first"], r["gemm_sum_ms"], b["gemm_sum_ms"],
        r["transform_ms"], b["transform_ms"], r["probe_mhz_mean"], b["probe_mhz_mean"], r["smu"]["sclk_mhz_mean"], r["smu"]["power_w_mean"],
        b["smu"]["sclk_mhz_mean"], b["smu"]["power_w_mean"], {True: "same", False: "DIFFER"}[r["bit_identical_to_breadth_first"]]))
if out_path:
    json.dump({"B": B, "passes_timed": NPASS, "warm_s": WARM_S, "rows": rows}, open(out_path, "w"), indent=1)
