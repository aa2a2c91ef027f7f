#!/usr/bin/env python
"""bench.py - keypoints/s of the YOHO hot path on MI355X.

A "step" is one pass of the hot path over one synthetic scene pair, inputs already resident
in HBM:  PartI descriptor on both fragments (2 x 5000 keypoints x 60 rotations x 32-D)
-> numpy-order invariant pooling -> mutual NN -> coarse rotation index -> PartII -> per-match
hypotheses -> YOHO-O vote (<=1000 hypotheses).  Whole-job keypoints/s = 10000 * pairs / time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; the checkpoint is broadcast once from rank 0 over RCCL, no data-path collective.
  --scaling weak (default): every rank runs its own pair per step (per-GPU work fixed);
  --scaling strong: a step is one sweep over a FIXED list of 64 pairs (8 synthetic scenes x 8 pairs) dealt to the ranks by
    yoho_amd.run_dataset.plan_shards (the dataset driver's plan); value = 64 * 10000 * steps / time.
Besides the headline (descriptor + YOHO-O) the line carries
  "sustained"  the same steps for >= 2 s behind >= 1 s of load, with the mean shader clock of probes spread over the region (the
               headline's 20 steps are a 0.1 s burst at boost clock);
  "yohoc"      the same step with the YOHO-C estimator (config 5: descriptor + 1000 RANSAC iterations, no PartII) in BOTH of its
               modes: sampled on the device, and the reference-exact host-parity mode (np.random draws + LAPACK sign on the host);
  "fcgf"       SURVEY 8(f) #3, the raw-cloud path: yoho_extractor.run on a seeded 300 k-point cloud (60 rotated backbone passes +
               feature transfer + PartI) -> ms per fragment with its phase split and the issued-MFMA fraction of the 3^3
               convolutions per level;
  "dataset"    the dataset-scale leg (60 fragments from disk, ~500 pairs);
  "roofline"   ONE definition: frac = fp16 MFMA flops the formulation needs (irrep GEMMs, 3 split products, no padding) / dense fp16
               peak over the four GEMM launches; frac_pass / frac_step the same flops over the whole PartI pass / the timed step;
  "cpu_baseline" the oracle timed on the host cores on the 5000-keypoint workload (one batch per network, scaled by row count).
Every leg reports the fp16 range guard's repeats (a checkpoint that trips it costs ~3x).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from yoho_amd import hip, synth, weights as W, pipeline, dist as ydist  # noqa: E402
from yoho_amd.power import PowerMonitor, ClockProbe  # noqa: E402

KP = 5000                       # keypoints per fragment (BASELINE.json configs[1])
DEFAULT_SCHEDULE = "0"          # PartI schedule of the timed steps (profiles/r03_chunk_sweep.md)
FLOP_PER_KP = 434_503_680       # 2 * 60 * (416*256 + 3328*512 + 6656*256 + 3328*32)  
FP32_MFMA_PEAK = 157.3          # TFLOP/s, MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32 = vector rate)
BF16_MFMA_PEAK = 2500.0         # TFLOP/s dense, MI355X_MICROARCH.md (v_mfma_f32_32x32x16_bf16)
BF16X3_EXEC_PER_ALG = 6.0 * 14.0 / 13.0   # bf16 MFMA flops issued per algorithmic flop: 6 cross products, 13 taps in 7 pairs
FOURIER_EXEC_PER_ALG = 244.0 / 780.0      # slab products per 8-channel chunk: sum_rho d^3 = 244 vs 60 x 13 = 780
FP16_MFMA_PEAK = 2500.0         # TFLOP/s dense (v_mfma_f32_32x32x16_f16, same rate as bf16)
NOMINAL_MHZ = 2400.0            # shader clock the peaks are quoted at (MI355X_MICROARCH.md)


def _issued_rows(d, cout, mode):
    """MFMA rows the GEMM kernel of `mode` issues for an irrep of dimension d: the live rows are d * cout; 256 x 256 tiles issue
    them in 128-row wave halves (a half made of padding only is skipped); the default mode's small-M kernel (fgemm3s, 32 output
    channels) issues exactly the 32 d live rows."""
    if mode == "fgemm" and cout == 32:
        return 32 * d
    return (d * cout + 127) // 128 * 128


def fgemm_issued_flops_per_layer(nkp, mode="fgemm"):
    """fp16 MFMA flops each of the four irrep-GEMM launches of one PartI pass over nkp keypoints issues, padding included:
    per irrep (d = 1,3,3,4,5) an (issued rows) x (N = d*kppad) x (K = d*Cin) product, 3 split products."""
    kppad = (nkp + 255) // 256 * 256
    return [sum(2 * 3 * _issued_rows(d, cout, mode) * (d * kppad) * (d * cin) for d in (1, 3, 3, 4, 5))
            for cin, cout in ((32, 256), (256, 512), (512, 256), (256, 32))]


def fgemm_issued_flops(nkp, mode="fgemm"):
    return sum(fgemm_issued_flops_per_layer(nkp, mode))


PMC_FILES = ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")     # newest first


def pmc_traffic(mode):
    """(HBM bytes per group-conv launch, provenance).  Average over the 4 launches of a PartI pass, measured with
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_traffic.py over tools/pmc_partI.py) and corrected
    as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950).  Counters cannot be read from inside a timed run, so the
    figure comes from the newest profiles/*_pmc_traffic.json, whose commit and pass totals are reported beside it."""
    for fn in PMC_FILES:
        path = os.path.join(REPO, "profiles", fn)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        per = d.get("partI_pass_gconv_bytes", {})
        if mode not in per:
            continue
        src = {"file": "profiles/" + fn, "commit": d.get("commit"), "pass_gconv_bytes": round(per[mode]),
               "pass_total_bytes": d.get("partI_pass_total_bytes", {}).get(mode)}
        return round(per[mode] / per.get("launches_per_pass", 4)), src
    return None, None


def measure_traffic(mode, nkp):
    """Fabric-side bytes of the irrep-GEMM launches MEASURED IN THIS RUN: two child processes of this command on this box run
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, nothing else traced, as MI355X_MICROARCH.md's HBM
    section prescribes) over tools/pmc_partI.py - two PartI passes of nkp keypoints in `mode` with the checkpoint of the timed steps -
    while this process waits.  gfx950 correction of the guide: FETCH_SIZE x 2 (16-byte-per-lane streaming reads are tallied at half
    their bytes; every operand of these kernels arrives by 16-byte LDS DMA), WRITE_SIZE as reported.  Both counters sit on the L2's
    memory side: Infinity-Cache hits are counted.  -> (dict or None, note)"""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import pmc_traffic
    tmp = tempfile.mkdtemp(prefix="yoho_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, PMC_B=str(nkp), TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        if any(k.startswith(("ROCPROF", "ROCP_")) or (k in ("HSA_TOOLS_LIB", "LD_PRELOAD") and "rocprof" in v) for k, v in env.items()):
            return None, "this process itself runs under rocprofv3: the counter passes are not nested (run bench.py unprofiled for `traffic`)"
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", cnt, "--output-format", "csv", "-d", os.path.join(tmp, cnt), "--",
                                sys.executable, os.path.join(REPO, "tools", "pmc_partI.py"), mode], env=env, cwd=tmp,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {cnt} exited with {r.returncode}: {r.stderr.decode(errors='replace')[-200:]}"
        F, Wr = pmc_traffic.load(os.path.join(tmp, "FETCH_SIZE"), "FETCH_SIZE"), pmc_traffic.load(os.path.join(tmp, "WRITE_SIZE"), "WRITE_SIZE")
        rows, gemm_b, gemm_n, all_b = [], 0.0, 0, 0.0
        for key, (n, fkib) in F.items():
            wn, wkib = Wr.get(key, [n, 0.0])
            b = (2.0 * fkib + wkib * n / max(wn, 1)) * 1024.0           # bytes over all n dispatches of this (kernel, grid)
            all_b += b
            if "fgemm" in key[0]:
                gemm_b += b
                gemm_n += n
                rows.append({"kernel": key[0], "grid": key[1], "dispatches": n, "fetch_x2_bytes_per_launch": round(2048.0 * fkib / n),
                             "write_bytes_per_launch": round(1024.0 * wkib / max(wn, 1))})
        if gemm_n == 0:
            return None, "no irrep-GEMM dispatch in the counter files"
        passes = gemm_n / 4.0
        return {"bytes_per_launch": round(gemm_b / gemm_n), "gemm_bytes_per_pass": round(gemm_b / passes), "pass_bytes": round(all_b / passes),
                "passes_profiled": passes, "per_kernel": rows}, "measured in this run"
    except Exception as e:                                            # the headline must survive a failure of this leg
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(pr, e0, e1, match, dr):
    """The oracle (`kind: "port"`: the reference's op sequence on torch-CPU kernels for the networks and Des2R, numpy for matcher and
    YOHO-O) timed on the host cores ON THE BENCHED WORKLOAD - 2 x 5000 keypoints - with the two networks sampled by batch:

      PartI    one batch of 900 rows, the reference's test_batch_size (tests/extractor.py:51-58), x 10000 / 900.  Rows are
               independent (utils/network.py:86-105 acts per keypoint), so the pass is 11.1 such batches.
      matcher  in full: numpy-order mean of both fragments + both 5000 x 5000 searches + the mutual check.
      Des2R    in full on the M matches.
      PartII   one batch of 1000 rows (its test_batch_size, tests/extractor.py:175-186), x M / 1000.
      [R|t] + YOHO-O  in full (M hypotheses built, 1000 voted).

    The stages after PartI take the GPU path's outputs for this pair (e0, e1, match, dr) as their inputs: the CPU legs are timed, not
    trusted - parity is the tests' job.  Thread count: best of a 3-point sweep on a 128-row PartI slice."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import yoho_oracle as orc
    from yoho_amd.tables import default_tables
    tb = default_tables()
    sd1 = W.synth_state_dict(W.PARTI_SPEC, 7)
    sd2 = W.synth_state_dict(W.PARTII_SPEC, 8)
    ncpu = os.cpu_count() or 1
    K = pr["feat0"].shape[0]
    M = int(match.shape[0])
    orc.partI_forward_torch(pr["feat0"][:16], sd1, tb.N)               # first-call costs of torch's CPU kernels stay out of the sweep
    sweep = {}
    for nt in sorted({min(ncpu, 16), min(ncpu, 64), ncpu}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        orc.partI_forward_torch(pr["feat0"][:128], sd1, tb.N)
        sweep[nt] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    st = {}

    def timed(name, fn):
        t0 = time.perf_counter()
        out = fn()
        st[name] = time.perf_counter() - t0
        return out
    timed("partI_batch900", lambda: orc.partI_forward_torch(pr["feat0"][:900], sd1, tb.N))
    i0, i1 = timed("mean", lambda: (orc.group_mean_np(e0), orc.group_mean_np(e1)))
    m_cpu = timed("matcher", lambda: orc.mutual_match(i0, i1))
    m0, m1 = match[:, 0], match[:, 1]
    timed("des2r", lambda: orc.des2r_torch(e1[m1], e0[m0], tb.P))
    nb2 = min(1000, M)
    q1 = timed("partII_batch1000", lambda: orc.partII_forward_torch(pr["feat1"][m1[:nb2]], pr["feat0"][m0[:nb2]], e1[m1[:nb2]], e0[m0[:nb2]],
                                                                     dr[:nb2], sd2, tb.N, tb.P))
    k0, k1 = pr["keys0"][m0], pr["keys1"][m1]
    q = np.concatenate([q1] * ((M + nb2 - 1) // nb2))[:M]               # timing input for the geometry stages (M rows)
    T = timed("hypotheses", lambda: orc.hyp_from_quat(q, dr, k0, k1, tb.R32))
    order = np.arange(M)
    np.random.RandomState(0).shuffle(order)
    timed("yohoo_vote", lambda: orc.yohoo_select(k0, k1, T, order, 0.09, 1000))
    partI_s = st["partI_batch900"] * (2 * K / 900.0)
    partII_s = st["partII_batch1000"] * (M / float(nb2))
    total = partI_s + st["mean"] + st["matcher"] + st["des2r"] + partII_s + st["hypotheses"] + st["yohoo_vote"]
    out = {"value": round(2 * K / total, 2), "unit": "keypoints/s", "cores": int(best), "kind": "port",
           "sample": f"the benched pair, 2 x {K} keypoints, {M} matches: PartI timed on one 900-row batch (test_batch_size) and scaled x {2 * K / 900.0:.2f} "
                     f"(rows are independent), matcher / Des2R / hypotheses / YOHO-O in full at {K} x {K} and M = {M}, PartII timed on one "
                     f"{nb2}-row batch and scaled x {M / float(nb2):.2f}; {sum(st.values()) + sum(sweep.values()):.1f} s of CPU work",
           "pair_s": round(total, 2), "cores_available": ncpu,
           "stage_s": {"partI_scaled": round(partI_s, 2), "partI_batch900": round(st["partI_batch900"], 3), "mean": round(st["mean"], 3),
                       "matcher_5000x5000": round(st["matcher"], 3), "des2r": round(st["des2r"], 3), "partII_scaled": round(partII_s, 2),
                       "partII_batch1000": round(st["partII_batch1000"], 3), "hypotheses": round(st["hypotheses"], 3),
                       "yohoo_vote": round(st["yohoo_vote"], 3)},
           "thread_sweep_partI_128rows_s": {str(k): round(v, 3) for k, v in sweep.items()},
           "matcher_agrees_with_gpu": bool(np.array_equal(m_cpu, match)),
           "versions": {"torch": torch.__version__, "numpy": np.__version__}}
    chk = os.path.join(REPO, "profiles", "r02_reference_vs_port.json")
    if os.path.exists(chk):
        # the port against the real reference in the build container, same inputs (tools/time_reference_vs_port.py)
        c = json.load(open(chk))
        out["reference_timing_check"] = {k: c[k] for k in ("keypoints_per_fragment", "cores", "reference_total_s", "port_total_s",
                                                            "port_over_reference_time", "outputs")}
        out["reference_timing_check"]["source"] = "profiles/r02_reference_vs_port.json (tools/time_reference_vs_port.py, build container)"
    return out


def fcgf_leg(ctx, dev, points=300000, nkpts=5000, runs=5):
    """SURVEY 8(f) #3 on the driver's line: yoho_extractor.run (simple_yoho/yoho_extract.py:57-77) on a seeded surface cloud -
    per group element: f64 rotation, voxelisation, FCGF backbone, NN feature transfer; then PartI - as wall ms per fragment
    (median of `runs` calls, host work and the CPU copies of the results included, as the API returns them), and one further call
    with the library's phase profile on for the split and the matrix-pipe figures of the 3^3 convolutions."""
    from yoho_amd.yoho_extract import yoho_extractor
    fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
    ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7}, "state_dict": fsd}
    ex = yoho_extractor(fcgf_ckpt=ck, yoho_ckpt=W.synth_state_dict(W.PARTI_SPEC, 7))
    lctx = ex.ctx
    pc = synth.surface_cloud(points, seed=1, extent=3.0)
    wall = []
    for rep in range(runs + 1):                            # the first call sizes the workspaces
        np.random.seed(rep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kpts, inv, eqv = ex.run(pc, voxel_size=0.025, nkpts=nkpts)
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
    timed_runs = sorted(wall[1:])
    # streamed over fragments (yoho_extractor.run_many: PartI and the result copy of a fragment on a tail lane while the backbone lanes
    # work on the next one): what a caller that describes many fragments pays per fragment
    streamed = []
    for rep in range(2):
        np.random.seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nstream = sum(1 for _ in ex.run_many([pc] * 6, voxel_size=0.025, nkpts=nkpts))
        torch.cuda.synchronize()
        streamed.append((time.perf_counter() - t0) * 1e3 / nstream)
    # one lane (every pass on the caller's stream, as up to round 5): the A/B of the two-lane pipeline, and the call the phase
    # profile is taken on (its spans are consecutive events on ONE stream)
    lanes_default = ex.lanes
    ex.lanes = 1
    wall1 = []
    for rep in range(3):
        np.random.seed(rep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ex.run(pc, voxel_size=0.025, nkpts=nkpts)
        torch.cuda.synchronize()
        wall1.append((time.perf_counter() - t0) * 1e3)
    lctx.phase_profile(True)
    np.random.seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ex.run(pc, voxel_size=0.025, nkpts=nkpts)
    torch.cuda.synchronize()
    prof_wall = (time.perf_counter() - t0) * 1e3
    ph = lctx.phase_read()
    lctx.phase_profile(False)
    ex.lanes = lanes_default
    # PartI of the 5000 keypoints on its own (HIP events)
    x = ex._last_group_feats
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lctx.partI_forward(x, want_inv=True)
    e0.record()
    for _ in range(3):
        lctx.partI_forward(x, want_inv=True, check_range=False)
    e1.record()
    torch.cuda.synchronize()
    partI_ms = e0.elapsed_time(e1) / 3
    ms = lambda *names: round(sum(ph[n]["ms"] for n in names), 3)
    conv_names = [n for n in ph if n.startswith(("conv", "strided", "transposed", "heads"))]
    levels = {}
    for l in range(4):
        d = ph[f"conv3x3_level{l}"]
        if d["ms"] > 0:
            tf = d["mfma_flops"] / (d["ms"] * 1e-3) / 1e12
            levels[f"level{l}"] = {"ms": round(d["ms"], 3), "launches": d["launches"], "issued_fp16_mfma_tflops": round(tf, 1),
                                   "issued_frac_of_fp16_peak": round(tf / FP16_MFMA_PEAK, 4)}
    device_ms = sum(v["ms"] for v in ph.values()) + partI_ms
    return {"metric": "ms per fragment from the raw cloud (yoho_extractor.run: 60 x (rotate, voxelise, FCGF backbone, NN transfer) + PartI)",
            "points": points, "keypoints": nkpts, "voxel_size": 0.025, "rotations_per_backbone_pass": ex.rot_batch,
            "ms_per_fragment": round(timed_runs[len(timed_runs) // 2], 2), "ms_per_fragment_all": [round(v, 2) for v in wall[1:]],
            "fragments_per_s": round(1e3 / timed_runs[len(timed_runs) // 2], 2),
            "ms_per_fragment_streamed": round(min(streamed), 2),
            "streamed_note": "yoho_extractor.run_many over 6 fragments (identical values to run(), tests/test_gpu_dropin.py): wall / fragments, the better of two runs",
            "lanes": {"backbone_lanes": lanes_default, "ms_per_fragment_one_lane": round(sorted(wall1)[1], 2),
                      "note": "lanes = (stream, library context) pairs the four backbone passes of a fragment alternate over: a pass's voxelisation and "
                              "coordinate / kernel maps are queued while the previous pass's convolutions run on the other lane; identical bits "
                              "(tests/test_gpu_dropin.py); split_ms / phases_ms / spconv_3x3_per_level below are taken on a ONE-lane call "
                              "(consecutive spans on one stream)"},
            "split_ms": {"voxelise_and_maps": ms("voxelise", "coordinate_maps", "kernel_maps"), "voxelise": ms("voxelise"),
                         "coordinate_maps": ms("coordinate_maps"), "kernel_maps": ms("kernel_maps"),
                         "convolutions": ms(*conv_names), "nn_feature_transfer": ms("nn_feature_transfer"), "partI": round(partI_ms, 3),
                         "device_total": round(device_ms, 2), "profiled_call_wall": round(prof_wall, 2),
                         "host_and_copies": round(prof_wall - device_ms, 2)},
            "phases_ms": {k: round(v["ms"], 3) for k, v in ph.items()},
            "spconv_3x3_per_level": levels,
            "issued_note": "issued = fp16 MFMA flops launched (every 32-row tile walks all 27 offsets x cin x cout, 3 split products: "
                           "about half of them multiply empty region cells) / time of the level's 3^3 stride-1 convolutions / 2.5 PFLOP/s",
            "data": "synthetic surface cloud (yoho_amd.synth.surface_cloud seed 1), random-init ResUNetBN2C (seeded)",
            "range_repeats": int(lctx.range_fallbacks)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-yohoc", action="store_true", help="skip the YOHO-C leg")
    ap.add_argument("--no-fgemm8", action="store_true", help="skip the leg with the opt-in fp8-correction arithmetic (gconv mode 'fgemm8')")
    ap.add_argument("--staged", action="store_true",
                    help="compose the estimator side of every pair from the staged entries in Python (pipeline.run_pair: tensor-library index "
                         "kernels between the stages) instead of one yoho_register_pair call per pair")
    ap.add_argument("--timed-only", action="store_true",
                    help="warm-up and the headline's timed regions only, then a short JSON line: the kernel trace of this command contains "
                         "nothing but what the timed steps launch (profiles/rNN_kernel_trace_bench_timed.md)")
    ap.add_argument("--in-flight", type=int, choices=[1, 2], default=2,
                    help="pairs in flight: 2 (default) queues the descriptor pass of the next pair on a second HIP stream before waiting for "
                         "the current pair's read-backs (pipeline.PairStreamer); 1 runs the pairs strictly one after the other")
    ap.add_argument("--gconv", choices=["f32", "bf16x3", "fourier", "fp16x2", "fgemm", "fgemm256", "fgemm128", "fgemm8"], default=os.environ.get("YOHO_GCONV", "fgemm"),
                    help="PartI group conv: group-Fourier domain (fp32 MFMA), direct fp32 MFMA, direct 3-way bf16 split MFMA, "
                         "or direct 2-way fp16 split MFMA")
    ap.add_argument("--partII", choices=["f32", "bf16x3", "fp16x2", "cgemm", "cgemm8"], default=os.environ.get("YOHO_PARTII", "fp16x2"),
                    help="arithmetic of the two PartII cone layers")
    ap.add_argument("--repeats", type=int, default=3, help="the timed region (exactly --steps steps) is run this many times; ms_per_step / value are "
                                                           "the median repeat, min / max are reported beside it")
    ap.add_argument("--partI-schedule", default=os.environ.get("YOHO_PARTI_CHUNK", DEFAULT_SCHEDULE),
                    help="PartI pass: '0' breadth-first, 'C' or 'CxS' depth-first over chunks of C keypoints on S (1 or 2) streams")
    ap.add_argument("--no-dataset", action="store_true", help="skip the dataset-scale leg (tools/bench_dataset.py: 60 fragments from disk, ~500 pairs)")
    ap.add_argument("--no-fcgf", action="store_true", help="skip the raw-cloud leg (yoho_extractor.run on a 300 k-point cloud)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run PMC measurement of roofline.traffic (two short rocprofv3 child processes)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the sustained leg (>= 1 s of load, then >= 2 s of timed steps)")
    args = ap.parse_args()
    sched = [int(v) for v in str(args.partI_schedule).lower().split("x")] + [1]
    sched_chunk, sched_streams = sched[0], sched[1]

    rank, world, local = ydist.init_from_env("nccl" if args.gpus > 1 else None)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    dev = local if world > 1 else 0
    torch.cuda.set_device(dev)

    ctx = hip.Context(dev)
    sd1 = W.synth_state_dict(W.PARTI_SPEC, 7) if rank == 0 else None
    sd2 = W.synth_state_dict(W.PARTII_SPEC, 8) if rank == 0 else None
    sd1 = ydist.broadcast_state_dict(sd1, W.PARTI_SPEC)       # RCCL broadcast, once
    sd2 = ydist.broadcast_state_dict(sd2, W.PARTII_SPEC)
    ctx.load_partI(sd1)
    ctx.load_partII(sd2)
    ctx.set_gconv_mode(args.gconv)
    ctx.set_partII_mode(args.partII)
    ctx.set_partI_schedule(sched_chunk, sched_streams)

    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.RandomState(1234 + rank)
    if args.scaling == "weak":
        # every rank owns a different synthetic pair (weak scaling: per-GPU work is fixed)
        pr = synth.make_pair(KP, seed=10 + rank)
        mine = [(cu(pr["feat0"]), cu(pr["feat1"]), cu(pr["keys0"]), cu(pr["keys1"]))]
        pairs_per_step = world
        full_plan = plan_cost = None
    else:
        # a fixed list of 64 pairs = 8 scenes x 8 pairs, dealt to the ranks by the dataset driver's plan (scenes whole, a
        # scene larger than a rank's share cut round-robin).  8 distinct synthetic pairs, each listed 8 times.
        from yoho_amd.run_dataset import plan_shards, plan_loads
        full_plan = plan_shards({f"scene{i}": 8 for i in range(8)}, world)
        plan_cost = plan_loads(full_plan)
        plan = full_plan[rank]
        distinct = {}
        mine = []
        for scene, positions in plan:
            for p in positions:
                sid = (int(scene[5:]) * 8 + p) % 8
                if sid not in distinct:
                    q = synth.make_pair(KP, seed=10 + sid)
                    distinct[sid] = (cu(q["feat0"]), cu(q["feat1"]), cu(q["keys0"]), cu(q["keys1"]))
                mine.append(distinct[sid])
        pairs_per_step = 64
    f0, f1, k0, k1 = mine[0]

    streamer = None
    if args.in_flight == 2:
        streamer = pipeline.PairStreamer(lambda: hip.Context(dev), sd1, sd2)
        streamer.set_modes(args.gconv, args.partII)
        for c_ in streamer.desc:
            c_.set_partI_schedule(sched_chunk, sched_streams)

    def run_steps(n, estimator="yohoo", seed0=0, hypotheses="all"):
        """n steps = n sweeps over this rank's pair list, all inside one call so that consecutive steps can overlap"""
        todo = [p for _ in range(n) for p in mine]
        dist = 0.09 if estimator == "yohoo" else 0.07
        if streamer is not None:
            # default: the estimator side of a pair is ONE library call (yoho_register_pair; vote order = RandomState(seed).shuffle restated
            # in C) - between the descriptor pass and the winner no tensor-library kernel runs (profiles/r06_kernel_trace_bench.md)
            return streamer.run(todo, inlier_dist=dist, max_iter=1000, order_rng=rng, estimator=estimator,
                                seeds=[seed0 + i for i in range(len(todo))], hypotheses=hypotheses, keep="last", fused=not args.staged)[-1]
        r = None
        for i, (a0, a1, b0, b1) in enumerate(todo):
            r = pipeline.run_pair(ctx, a0, a1, b0, b1, inlier_dist=dist, max_iter=1000, order_rng=rng, estimator=estimator, seed=seed0 + i,
                                  hypotheses=hypotheses)
        return r

    mon = PowerMonitor(dev)
    if os.environ.get("YOHO_BENCH_SMU", "1") == "0":
        mon.source = None                                  # no SMU sampling thread (diagnostics)
    PROBE_US = int(os.environ.get("YOHO_BENCH_PROBE_US", "20"))       # 0 switches the device-side clock probes off
    PROBES_PER_STEP = float(os.environ.get("YOHO_BENCH_PROBES_PER_STEP", "4"))

    class _NoProbe:
        def queue(self, n=1):
            pass

        def summary(self):
            return None
    probe = ClockProbe(ctx, us=PROBE_US) if PROBE_US > 0 else _NoProbe()

    def timed(estimator, steps, warmup, repeats=1, hypotheses="all"):
        """warmup steps, then `repeats` timed regions of exactly `steps` steps each (barrier + synchronize on both sides, max over
        ranks) -> (median region time, all region times, per-rank times of the median region, last result, clock / power during
        the regions)"""
        # The SMU is read ONCE, right behind the last timed region: any amdsmi access while the regions run - a side-thread poller, or
        # a single read between two regions - costs one of the following regions 35-45 ms (measured: 5.39 / 7.19 / 5.43 ms per step
        # with a poller, 5.42 / 5.48 / 7.09 with a read between regions, 5.41 / 5.40 / 5.39 with none), and the socket-power figure
        # is a ~1 s moving average anyway, i.e. still the regions'; the clock DURING the regions comes from the device-side probe
        probe.queue(1)                                     # the probe stream's creation happens during the warm-up
        r = run_steps(warmup, estimator, 1, hypotheses)
        probe.summary()
        times, per_rank, smu_after = [], [], []
        import gc
        for rep in range(max(1, repeats)):
            gc.collect()                                   # the interpreter's collector stays out of the timed region: a stalled region
            gc.disable()                                   # (6.9 ms per step beside 5.41 / 5.41) showed the device idle, i.e. the host late
            ydist.barrier()
            torch.cuda.synchronize()
            probe.queue(max(1, int(steps * PROBES_PER_STEP)))            # short probes on their own high-priority stream
            t0 = time.perf_counter()
            r = run_steps(steps, estimator, 1000 + 100000 * rep, hypotheses)
            torch.cuda.synchronize()
            mine_dt = time.perf_counter() - t0
            ydist.barrier()
            times.append(ydist.max_over_ranks(time.perf_counter() - t0))
            per_rank.append(ydist.all_ranks(mine_dt))
            gc.enable()
        smu_after.append(mon.read_once())
        power = {"power_w_after_last_region": smu_after[0]["power_w"], "power_cap_w": None if mon.cap_w is None else round(float(mon.cap_w), 1),
                 "source": mon.source, "clock_probe": probe.summary()}
        order = sorted(range(len(times)), key=lambda i: times[i])
        med = order[len(order) // 2]
        return times[med], times, per_rank[med], r, power

    def guard_total():
        return int(ctx.range_fallbacks + (sum(c.range_fallbacks for c in streamer.desc + [streamer.est]) if streamer else 0))

    g0 = guard_total()
    dt, dts, rank_dts, res_timed, power_steps = timed("yohoo", args.steps, max(args.warmup, 1), args.repeats)
    headline_range_repeats = guard_total() - g0
    if args.timed_only:
        if rank == 0:
            print(json.dumps({"metric": "keypoints/sec (5000 kp x60 rot desc+YOHO-O)", "value": round(pairs_per_step * 2 * KP * args.steps / dt, 1),
                              "unit": "keypoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                              "timed_only": True, "staged": bool(args.staged), "matches": int(res_timed.matches), "winner": int(res_timed.best_h),
                              "range_repeats": headline_range_repeats}), flush=True)
        ydist.barrier()
        return
    # the last timed pair once more through the staged entries (untimed) with the vote order of its seed: the stage outputs the legs
    # below need (match list, coarse rotations, descriptors), and the check that the one-call pair of the timed steps picked the same
    # winner and transform
    last_seed = 1000 + 100000 * (max(1, args.repeats) - 1) + args.steps * len(mine) - 1
    res = pipeline.run_pair(ctx, *mine[-1], inlier_dist=0.09, max_iter=1000, order_rng=np.random.RandomState(last_seed & 0xFFFFFFFF))
    one_call = {"estimator_side": "staged entries composed in Python (pipeline.run_pair)" if (args.staged or streamer is None) else
                                  "one library call per pair (yoho_register_pair)",
                "matches_winner_count_transform_equal_staged_composition":
                    bool(res_timed.matches == res.matches and (res_timed.best_h, res_timed.best_count) == (res.best_h, res.best_count) and
                         np.array_equal(np.asarray(res_timed.trans), np.asarray(res.trans)))}

    # sustained leg: the headline's timed region is 0.1 s behind a few warm-up steps, i.e. a burst at boost clock.  Here the same
    # steps run for >= 1 s untimed (the part reaches the clock its power budget allows) and then >= 2 s timed, with clock probes
    # spread over the whole timed region by a side thread (one 20 us probe every ~10 ms on its own context and stream).
    sustained = None
    if not args.no_sustained:
        import threading
        step_s = dt / args.steps
        n_load, n_timed = max(args.steps, int(np.ceil(1.0 / step_s))), max(args.steps, int(np.ceil(2.0 / step_s)))
        pctx = hip.Context(dev) if PROBE_US > 0 else None
        sprobe = ClockProbe(pctx, us=PROBE_US, capacity=1024) if pctx is not None else None
        stop = threading.Event()

        def prober():
            torch.cuda.set_device(dev)
            while not stop.is_set():
                sprobe.queue(1)
                stop.wait(0.010)
        g1 = guard_total()
        import gc
        gc.collect()
        gc.disable()                                       # as in timed(): the interpreter's collector stays out of the region
        run_steps(n_load, "yohoo", 7)
        ydist.barrier()
        torch.cuda.synchronize()
        th = threading.Thread(target=prober, daemon=True) if sprobe is not None else None
        if th is not None:
            th.start()
        t0 = time.perf_counter()
        run_steps(n_timed, "yohoo", 8)
        torch.cuda.synchronize()
        sdt_mine = time.perf_counter() - t0
        stop.set()
        if th is not None:
            th.join()
        gc.enable()
        ydist.barrier()
        sdt = ydist.max_over_ranks(sdt_mine)
        sustained = {"ms_per_step": round(sdt / n_timed * 1e3, 3), "value": round(pairs_per_step * 2 * KP * n_timed / sdt, 1), "unit": "keypoints/s",
                     "load_steps_before": n_load, "timed_steps": n_timed, "timed_s": round(sdt, 3),
                     "clock_probe": sprobe.summary() if sprobe is not None else None, "power_w_after": mon.read_once()["power_w"],
                     "vs_headline_ms_per_step": round((sdt / n_timed) / (dt / args.steps), 4), "range_repeats": guard_total() - g1,
                     "note": "the same step as the headline, >= 1 s of untimed load then >= 2 s timed; clock_probe = 20 us one-wave probes every "
                             "~10 ms over the whole timed region (mean = the clock the part sustains under this load; 2400 MHz nominal)"}
    # the same step with PartII evaluated only for the 1000 matches the YOHO-O vote reads (pipeline.run_pair hypotheses="selected":
    # identical winner / transform, tests/test_gpu_fullsize.py); reported beside the headline, which keeps the reference's
    # workload (PartII and [R|t] for every match, as its Trans_pre stage leaves them)
    sel_leg = None
    if not args.no_yohoc:
        g1 = guard_total()
        dts_, _, _, ress, _ = timed("yohoo", args.steps, max(min(args.warmup, 2), 1), 1, "selected")
        sel_leg = {"metric": "keypoints/sec (5000 kp x60 rot desc + YOHO-O, PartII only for the 1000 voted hypotheses)",
                   "value": round(pairs_per_step * 2 * KP * args.steps / dts_, 1), "ms_per_step": round(dts_ / args.steps * 1e3, 3),
                   "winner_inliers": int(ress.best_count), "range_repeats": guard_total() - g1}
        # one pair both ways with the same shuffle: the winner and the transform must be the same
        ra_ = pipeline.run_pair(ctx, f0, f1, k0, k1, max_iter=1000, order_rng=np.random.RandomState(7))
        rs_ = pipeline.run_pair(ctx, f0, f1, k0, k1, max_iter=1000, order_rng=np.random.RandomState(7), eqv=ra_.eqv, hypotheses="selected")
        sel_leg["same_winner_and_transform_as_all"] = bool((ra_.best_h, ra_.best_count) == (rs_.best_h, rs_.best_count) and
                                                           np.array_equal(np.asarray(ra_.trans), np.asarray(rs_.trans)))
    # the same step in the opt-in arithmetic with EVERY large split product's corrections on the fp8 pipe (v_mfma_scale_f32_32x32x64_f8f6f4):
    # gconv mode 'fgemm8' (round 5: the two large PartI layers, NOTEBOOK.md 3.1h) + PartII mode 'cgemm8' (round 6: the 13-element cone layer
    # as an implicit GEMM with fp8 corrections); also PartII 'cgemm8' alone beside the default PartI.  Time, and what it does to
    # this pair's results.  Reported beside the headline, which stays on the default arithmetic (~1e-6 of the fp32 reference; these ~1e-5,
    # tolerance 1e-4; tests/test_gpu_census.py counts what that costs against the reference itself)
    fgemm8 = None
    if not args.no_fgemm8 and args.gconv == "fgemm" and args.partII == "fp16x2":
        g1 = guard_total()

        def leg(gm, pm):
            ctx.set_gconv_mode(gm)
            ctx.set_partII_mode(pm)
            if streamer is not None:
                streamer.set_modes(gm, pm)
            try:
                dt_, _, _, _, pw_ = timed("yohoo", args.steps, max(min(args.warmup, 2), 1), 1)
                ra_ = pipeline.run_pair(ctx, f0, f1, k0, k1, max_iter=1000, order_rng=np.random.RandomState(7))
            finally:
                ctx.set_gconv_mode(args.gconv)
                ctx.set_partII_mode(args.partII)
                if streamer is not None:
                    streamer.set_modes(args.gconv, args.partII)
            return dt_, pw_, ra_
        dt8, pw8, ra8 = leg("fgemm8", "cgemm8")
        dtp, _, rap = leg(args.gconv, "cgemm8")
        rd8 = pipeline.run_pair(ctx, f0, f1, k0, k1, max_iter=1000, order_rng=np.random.RandomState(7))
        e8 = torch.cat([ra8.eqv[0]["eqv"], ra8.eqv[1]["eqv"]])
        ed = torch.cat([rd8.eqv[0]["eqv"], rd8.eqv[1]["eqv"]])
        same_list = bool(ra8.match.shape == rd8.match.shape and torch.equal(ra8.match, rd8.match))
        fgemm8 = {"metric": "keypoints/sec (5000 kp x60 rot desc + YOHO-O), every large split product's corrections in fp8 (opt-in: gconv 'fgemm8' + PartII 'cgemm8')",
                  "value": round(pairs_per_step * 2 * KP * args.steps / dt8, 1), "ms_per_step": round(dt8 / args.steps * 1e3, 3),
                  "vs_headline_ms_per_step": round(dt8 / dt, 4),
                  "descriptor_max_abs_diff_vs_default": float((e8 - ed).abs().max().item()),
                  "quaternion_max_abs_diff_vs_default": float((ra8.quat - rd8.quat).abs().max().item()) if same_list else None,
                  "same_match_list_as_default": same_list,
                  "same_winner_as_default": bool((ra8.best_h, ra8.best_count) == (rd8.best_h, rd8.best_count)),
                  "matches": int(ra8.match.shape[0]), "clock_probe": (pw8 or {}).get("clock_probe"),
                  "partII_cgemm8_only": {"ms_per_step": round(dtp / args.steps * 1e3, 3), "vs_headline_ms_per_step": round(dtp / dt, 4),
                                         "quaternion_max_abs_diff_vs_default": float((rap.quat - rd8.quat).abs().max().item()),
                                         "same_winner_as_default": bool((rap.best_h, rap.best_count) == (rd8.best_h, rd8.best_count))},
                  "range_repeats": guard_total() - g1}
    yohoc = None
    if not args.no_yohoc:
        g1 = guard_total()
        dtc, _, _, resc, _ = timed("yohoc", args.steps, max(min(args.warmup, 2), 1))
        # host time per pair of the estimator call alone (launches only: nothing is read back inside the call)
        m_, dr_ = res.match, res.dr_index
        torch.cuda.synchronize()
        th = time.perf_counter()
        for i in range(20):
            ctx.c_ransac_device(k0, k1, dr_, 1000, 100 + i, 0.07, match=m_)
        host_ms = (time.perf_counter() - th) / 20 * 1e3
        torch.cuda.synchronize()
        # the estimator call alone, device mode, start to result on the host
        est_dev = []
        for i in range(5):
            torch.cuda.synchronize()
            th = time.perf_counter()
            T_, r_, _ = ctx.c_ransac_device(k0, k1, dr_, 1000, 200 + i, 0.07, match=m_)
            torch.cat([T_.reshape(-1), r_.to(torch.float64)]).cpu()
            est_dev.append((time.perf_counter() - th) * 1e3)
        # the reference-exact mode of the drop-in class (yoho_amd.estimator.yohoc, default): np.random draws consumed as
        # tests/estimator.py:119-128 does + one batched np.linalg.svd for LAPACK's reflection sign on the HOST, Kabsch + vote for the
        # 1000 iterations in one device call - same matches, same coarse rotations, per pair
        import types as _types
        from yoho_amd import estimator as yest
        yc = yest.yohoc(_types.SimpleNamespace(ransac_c_inlinerdist=0.07, SO3_related_files=None))
        mh = m_.cpu().numpy()
        km0, km1, drh = k0.cpu().numpy()[mh[:, 0]], k1.cpu().numpy()[mh[:, 1]], dr_.cpu().numpy()
        np.random.seed(99)
        yc.estimate_host_sampled(km0, km1, drh, 1000)                       # warm
        tm, est_host = {}, []
        for i in range(3):
            th = time.perf_counter()
            yc.estimate_host_sampled(km0, km1, drh, 1000, timings=tm)
            est_host.append((time.perf_counter() - th) * 1e3)
        yohoc = {"metric": "keypoints/sec (5000 kp x60 rot desc + YOHO-C, 1000 iterations sampled on the device)",
                 "value": round(pairs_per_step * 2 * KP * args.steps / dtc, 1), "ms_per_step": round(dtc / args.steps * 1e3, 3),
                 "iterations": 1000, "estimator_host_ms_per_pair": round(host_ms, 4),
                 "winner_inliers": int(resc.best_count), "matches": int(resc.matches), "range_repeats": guard_total() - g1,
                 "modes": {
                     "device_sampling": {"contract": "statistical parity: Philox-sampled triples, proper rotations, no host work (pipeline.run_pair, "
                                                     "run_dataset, cfg.yohoc_device_sampling); bit-exact vs oracle/yoho_oracle.yohoc_device_triples",
                                         "estimator_ms_per_pair": round(float(np.median(est_dev)), 3),
                                         "host_ms_per_pair": round(host_ms, 4), "step_ms": round(dtc / args.steps * 1e3, 3)},
                     "host_parity": {"contract": "reference-exact (yoho_amd.estimator.yohoc default): np.random stream consumed as tests/estimator.py:119-128, "
                                                 "LAPACK reflection sign from one batched np.linalg.svd on the host, Kabsch + vote on the device",
                                     "estimator_ms_per_pair": round(float(np.median(est_host)), 3),
                                     "of_which_np_random_draws_ms": round(tm["draw_s"] / 3 * 1e3, 3),
                                     "of_which_svd_sign_mask_ms": round(tm["svd_mask_s"] / 3 * 1e3, 3),
                                     "of_which_device_call_ms": round(tm["device_call_s"] / 3 * 1e3, 3),
                                     "note": "the cost of LAPACK / np.random exactness is host time: the device call is the same Kabsch + vote kernel; "
                                             "a step in this mode = descriptor pass + matcher + Des2R + this (not overlapped: the draws hold the GIL)"}}}

    # per-kernel timing of the dominant kernel (group conv), HIP events on the launch stream; same batch as the
    # timed step (both fragments in one pass)
    fboth = torch.cat([f0, f1])
    nkp = fboth.shape[0]
    # (with two chunk streams the launches of neighbouring chunks overlap and their event times would count the overlap twice, so
    # the per-launch figures come from the same chunks on ONE stream; the whole-pass time of the schedule as timed is reported too)
    ctx.set_partI_schedule(sched_chunk, 1)
    ctx.set_profiling(True)
    conv_ms = []
    for _ in range(3):
        # eight passes back to back, then the ninth - queued behind them with no idle gap - is the one whose events are read: the
        # launches are timed at the clock the part holds under sustained PartI load, as in the timed steps, not after a pause
        for _ in range(9):
            ctx.partI_forward(fboth, want_inv=False, want_inv_np=True, check_range=False)
        probe.queue(4)
        torch.cuda.synchronize()
        conv_ms.append([ctx.kernel_ms(i) for i in range(13)])
    power_prof = {"power_w_after": mon.read_once()["power_w"], "clock_probe": probe.summary()}
    pass_ms_timed_schedule = None
    if sched_streams == 2:
        ctx.set_partI_schedule(sched_chunk, 2)
        t_ = []
        for _ in range(3):
            ctx.partI_forward(fboth, want_inv=False, want_inv_np=True, check_range=False)
            torch.cuda.synchronize()
            t_.append(ctx.kernel_ms(12))
        pass_ms_timed_schedule = float(np.mean(t_))
    ctx.set_profiling(False)
    ctx.set_partI_schedule(sched_chunk, sched_streams)
    conv_ms = np.array(conv_ms).mean(0)
    gconv_total_ms = float(conv_ms[:4].sum())
    achieved = FLOP_PER_KP * nkp / (gconv_total_ms * 1e-3) / 1e12      # algorithmic (direct 13-tap) FLOP/s
    conv_total_ms = float(conv_ms[:4].sum() + conv_ms[4] + conv_ms[5] + conv_ms[6])    # GEMMs + head + tail + transforms

    # the HBM-bound kernels of a step, each against its ALGORITHMIC bytes (what the kernel must read and write once)
    def ev_ms(fn, n=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    o0, o1 = res.eqv
    M = int(res.match.shape[0])
    nn_ms = ev_ms(lambda: ctx.mutual_nn(o0["inv_np"], o1["inv_np"]))
    des_ms = ev_ms(lambda: ctx.des2r_matched(o1["eqv"], o0["eqv"], res.match))
    coef = lambda ch: 60 * ch * 4 * nkp                  # bytes of one fp32 coefficient (or fp16x2 plane) tensor with ch channels
    hbm = {}

    def add(name, ms, nbytes, note):
        hbm[name] = {"ms": round(float(ms), 4), "bytes": int(nbytes), "TBps": round(nbytes / (ms * 1e-3) / 1e12, 3) if ms > 0 else None,
                     "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 8e12, 4) if ms > 0 else None, "what": note}
    if args.gconv in ("fgemm", "fgemm256", "fgemm128", "fgemm8"):
        add("head16_kernel", conv_ms[4], 2 * coef(32), "x (B,32,60) f32 in, cin=32 operand planes out")
        for i, ch in ((7, 256), (8, 512), (9, 256)):
            add(f"gft16_kernel<ACTP> {ch}ch", conv_ms[i], 2 * coef(ch), "fp32 coefficients in, BN+ReLU in the group domain, fp16x2 operand planes out")
        add("gft16_kernel<INV>", conv_ms[10], 2 * coef(32), "fp32 coefficients in, group-domain fp32 out")
        add("finalize_partI_kernel", conv_ms[11], 3 * coef(32) + 128 * nkp, "y + x in, eqv + inv_np out")
    if os.environ.get("YOHO_NN") == "brute":
        add("nn32seg + mutual_compact (both directions, brute force)", nn_ms, 4 * KP * 128 + 16 * M,
            "vector-fp32-bound, not HBM-bound: 2 x 5000 x 5000 x 32 x 3 flop = %.1f TFLOP/s of the 157.3 fp32 peak" % (2 * KP * KP * 32 * 3 / (nn_ms * 1e-3) / 1e12))
    else:
        add("mutual NN: mf_norms + 2 x mf_gram (fp16-MFMA Gram pre-filter, exact fp32 distance of the candidates in the error band) + mutual_compact",
            nn_ms, 4 * KP * 128 + 16 * M,
            "latency / issue-bound, not HBM-bound: the %d x %d Gram matrix is formed twice on the matrix cores (row minima, then candidates), "
            "%.1f us per pass; brute force (YOHO_NN=brute) evaluates 2 x 5000 x 5000 explicit differences on the vector pipes" % (KP, KP, nn_ms * 500))
    add("des2r_kernel", des_ms, M * (2 * 7680 + 8), "two (32,60) descriptors per match in, index out")

    traffic_meas, traffic_note = (None, "not measured (--no-traffic, or more than one rank)")
    if rank == 0 and world == 1 and not args.no_traffic:
        torch.cuda.synchronize()
        traffic_meas, traffic_note = measure_traffic(args.gconv, nkp)
    if rank == 0:
        M = int(res.match.shape[0])
        if args.gconv == "f32":
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK, 4), "traffic": None, "traffic_replayed": pmc_traffic("f32")[0],
                    "kernel": "gconv_kernel<15,false> (4 launches = 4 PartI layers, 2.1725 algorithmic TFLOP per 5000 kp)"}
            dtype = "f32"
        elif args.gconv == "bf16x3":
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / BF16_MFMA_PEAK, 4), "traffic": None, "traffic_replayed": pmc_traffic("bf16x3")[0],
                    "kernel": "gconv16_kernel<15,2> + <8,1> (4 launches = 4 PartI layers, 2.1725 algorithmic TFLOP per 5000 kp)",
                    "executed_tflops": round(achieved * BF16X3_EXEC_PER_ALG, 1),
                    "executed_frac": round(achieved * BF16X3_EXEC_PER_ALG / BF16_MFMA_PEAK, 4),
                    "note": "achieved = algorithmic fp32-equivalent FLOP/s; the fp32-accurate bf16 split issues 6.46 bf16 MFMA flops "
                            "per algorithmic flop, so frac <= 0.155 for this formulation"}
            dtype = "bf16x3 split (fp32-accurate, fp32 accumulate)"
        elif args.gconv == "fp16x2":
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / BF16_MFMA_PEAK, 4), "traffic": None, "traffic_replayed": pmc_traffic("fp16x2")[0],
                    "kernel": "gconv16_kernel<15,2,2> + <8,1,2> (4 launches = 4 PartI layers, 2.1725 algorithmic TFLOP per 5000 kp)",
                    "executed_tflops": round(achieved * BF16X3_EXEC_PER_ALG / 2, 1),
                    "executed_frac": round(achieved * BF16X3_EXEC_PER_ALG / 2 / BF16_MFMA_PEAK, 4),
                    "note": "achieved = algorithmic fp32-equivalent FLOP/s; the 2-way fp16 split (x = hi + lo, 3 products, error "
                            "<= 3*2^-22 per product) issues 3.23 fp16 MFMA flops per algorithmic flop, so frac <= 0.31"}
            dtype = "fp16x2 split (2^-22-accurate products, fp32 accumulate)"
        elif args.gconv in ("fgemm", "fgemm256", "fgemm128", "fgemm8"):
            # ONE definition: the fp16 MFMA flops the formulation needs - the irrep GEMMs (244 / 780 of the direct
            # multiply-adds), every product as 3 fp16 MFMA products, NO padding rows or columns - over the dense fp16 peak
            useful = FLOP_PER_KP * nkp * FOURIER_EXEC_PER_ALG * 3.0
            useful_layer = [2.0 * 244 * cin * cout * nkp * 3.0 for cin, cout in ((32, 256), (256, 512), (512, 256), (256, 32))]
            issued = fgemm_issued_flops(nkp, "fgemm" if args.gconv == "fgemm8" else args.gconv) / (gconv_total_ms * 1e-3) / 1e12
            pass_ms = float(conv_ms[12])
            step_ms = dt / args.steps * 1e3 / (len(mine) if args.scaling == "strong" else 1)
            ach = useful / (gconv_total_ms * 1e-3) / 1e12
            traffic, tsrc = pmc_traffic("fgemm" if args.gconv == "fgemm8" else args.gconv)
            # algorithmic bytes of the four GEMM launches of one pass: fp16x2 operand planes in (4 B per value), fp32 coefficients out,
            # the residual of the third layer in, every weight pack once (244 / 13 of the 13-tap weights, 4 B per value)
            alg_bytes = sum(60 * 4 * nkp * (cin + cout) + 244 * 4 * cin * cout for cin, cout in ((32, 256), (256, 512), (512, 256), (256, 32))) + 60 * 4 * nkp * 256
            # the box's state beside the fraction (a slow box and a slow build must be told apart from this one object):
            # the clock the GEMM launches of the profiled passes ran at, and the fraction against the peak AT THAT CLOCK
            cp_prof = power_prof.get("clock_probe") or {}
            cp_step = power_steps.get("clock_probe") or {}
            mhz_prof = cp_prof.get("shader_mhz_mean")
            mhz_step = cp_step.get("shader_mhz_mean")
            at_clock = lambda frac, mhz: round(frac / (mhz / NOMINAL_MHZ), 4) if mhz else None
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP16_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(ach / FP16_MFMA_PEAK, 4),
                    "shader_mhz_mean": mhz_prof, "shader_mhz_nominal": NOMINAL_MHZ,
                    "frac_at_clock": at_clock(ach / FP16_MFMA_PEAK, mhz_prof),
                    "power_w": power_prof.get("power_w_after"), "power_cap_w": power_steps.get("power_cap_w"),
                    "timed_steps": {"shader_mhz_mean": mhz_step, "power_w": power_steps.get("power_w_after_last_region"),
                                    "frac_step_at_clock": at_clock(useful / (step_ms * 1e-3) / 1e12 / FP16_MFMA_PEAK, mhz_step)},
                    "clock_note": "shader_mhz_mean = mean of the one-wave clock probes queued with the profiled PartI passes `achieved` is measured on "
                                  "(timed_steps: with the timed steps); frac_at_clock = achieved / (peak x shader_mhz_mean / 2400): the fraction of what the "
                                  "matrix cores could do at the clock the power limit left them - comparable across boxes, `frac` is not",
                    "frac_pass": round(useful / (pass_ms * 1e-3) / 1e12 / FP16_MFMA_PEAK, 4),
                    "frac_step": round(useful / (step_ms * 1e-3) / 1e12 / FP16_MFMA_PEAK, 4),
                    "definition": "achieved = useful issued fp16 MFMA flops / time of the four GEMM launches of one PartI pass over both fragments; "
                                  "useful = 4.345 TFLOP (direct 13-tap count, SURVEY 8d) x 244/780 (irrep GEMMs) x 3 (fp16x2 split products) = "
                                  f"{useful / 1e12:.3f} TFLOP per 10000 keypoints, no padding; frac_pass = the same flops over the whole PartI pass "
                                  f"(GEMMs + transforms + head + tail, {pass_ms:.3f} ms), frac_step over the timed step ({step_ms:.3f} ms)",
                    "frac_per_launch": [round(f / (ms * 1e-3) / 1e12 / FP16_MFMA_PEAK, 4) for f, ms in zip(useful_layer, conv_ms[:4])],
                    "launch_ms_sum": round(gconv_total_ms, 4), "pass_ms": round(pass_ms, 4), "step_ms": round(step_ms, 4),
                    # fabric-side bytes per GEMM launch (average of the four launches of a pass), MEASURED IN THIS RUN by two rocprofv3 child
                    # processes (measure_traffic); null only when that failed (traffic_note says why)
                    "traffic": None if traffic_meas is None else traffic_meas["bytes_per_launch"],
                    "traffic_note": traffic_note,
                    "traffic_detail": None if traffic_meas is None else dict(
                        traffic_meas, algorithmic_bytes_per_launch=round(alg_bytes / 4.0),
                        over_algorithmic=round(traffic_meas["gemm_bytes_per_pass"] / alg_bytes, 2),
                        definition="(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch from two separate `rocprofv3 --kernel-trace --pmc` passes over "
                                   "tools/pmc_partI.py (two PartI passes of the same keypoint count, same checkpoint, same arithmetic mode) run as child "
                                   "processes of this command; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note; the counters sit on the L2's "
                                   "memory side and count Infinity-Cache hits: the excess over `algorithmic` is the weight panels every column tile "
                                   "re-streams from the Infinity Cache (31 MB of packs, on die), the activation planes and the coefficient writes are "
                                   "at their algorithmic volume (profiles/r06_nt_traffic.md); algorithmic = operand planes in + fp32 coefficients out + "
                                   "residual in + weight packs once, summed over the four launches / 4"),
                    "traffic_replayed": dict(tsrc or {}, bytes_per_launch=traffic,
                                             note="the committed PMC file of an earlier run of the same command (tools/collect_profiles.sh), kept for comparison"),
                    "kernel": {"fgemm": "fgemm3_kernel (fgemm3s_kernel for the 32-channel layer)", "fgemm128": "fgemm2_kernel", "fgemm256": "fgemm_kernel",
                               "fgemm8": "fgemm3c_kernel for 256 -> 512 and 512 -> 256 (fp16 main product + fp8 e4m3 corrections: 2 / 3 of the matrix time the `useful` count assumes), fgemm3 / fgemm3s for the 32-channel layers"}[args.gconv] +
                              " (4 launches = 4 PartI layers over both fragments)",
                    "issued_with_padding": {"tflops": round(issued, 1), "frac": round(issued / FP16_MFMA_PEAK, 4),
                                            "frac_per_launch": [round(f / (ms * 1e-3) / 1e12 / FP16_MFMA_PEAK, 4)
                                                                for f, ms in zip(fgemm_issued_flops_per_layer(nkp, "fgemm" if args.gconv == "fgemm8" else args.gconv), conv_ms[:4])]},
                    "direct_form": {"tflops": round(achieved, 2), "frac_of_fp16_peak": round(achieved / FP16_MFMA_PEAK, 4),
                                    "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK, 3),
                                    "note": "the reference's direct 13-tap flop count (4.345 TFLOP per 10000 kp) over the same launches: what rounds 1-3 "
                                            "reported as `frac`; not a hardware utilisation (the kernel does 244/780 of those multiply-adds, as 3 products each)"},
                    "conv_total_ms": round(conv_total_ms, 3)}
            dtype = "fp16x2 split (2^-22-accurate products, fp32 accumulate)"
        else:
            ex = achieved * FOURIER_EXEC_PER_ALG
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK, 4), "traffic": None, "traffic_replayed": pmc_traffic("fourier")[0],
                    "kernel": "gconvf_kernel (4 launches = 4 PartI layers, 2.1725 algorithmic TFLOP per 5000 kp)",
                    "executed_tflops": round(ex, 2), "executed_frac": round(ex / FP32_MFMA_PEAK, 4),
                    "note": "achieved = algorithmic FLOP/s of the reference's direct 13-tap formulation (SURVEY 8d) over the 4 "
                            "gconvf launches; the kernel evaluates the same convolution on group-Fourier coefficients and issues "
                            "244/780 of those flops on v_mfma_f32_32x32x2_f32, so frac can exceed 1; executed_frac is the "
                            "fraction of the fp32-MFMA peak actually sustained. The transform kernels between the layers are "
                            "timed separately (roofline_extra.transform_ms)"}
            dtype = "f32"
        out = {
            "metric": "keypoints/sec (5000 kp x60 rot desc+YOHO-O)",
            "value": round(pairs_per_step * 2 * KP * args.steps / dt, 1),
            "unit": "keypoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "ms_per_step_repeats": {"n": len(dts), "min": round(min(dts) / args.steps * 1e3, 3), "median": round(dt / args.steps * 1e3, 3),
                                    "max": round(max(dts) / args.steps * 1e3, 3),
                                    "all_in_order": [round(v / args.steps * 1e3, 3) for v in dts],
                                    "note": "every repeat times exactly --steps steps between barriers; value / ms_per_step are the median repeat"},
            "ranks": {"world_size_seen": world, "process_group": bool(ydist.active()),
                      "backend": (torch.distributed.get_backend() if ydist.active() else None),
                      "ms_per_step_per_rank": {"min": round(min(rank_dts) / args.steps * 1e3, 3), "mean": round(float(np.mean(rank_dts)) / args.steps * 1e3, 3),
                                               "max": round(max(rank_dts) / args.steps * 1e3, 3), "all": [round(v / args.steps * 1e3, 3) for v in rank_dts]},
                      # what every rank was given and how long its share took (one step = one sweep over its pairs); under --scaling strong the
                      # shares come from run_dataset.plan_shards, whose predicted max / mean load is printed beside the measured one
                      "pairs_per_rank": [sum(len(pos) for _, pos in part) for part in full_plan] if full_plan else [1] * world,
                      "ms_per_rank": [round(v / args.steps * 1e3, 3) for v in rank_dts],
                      "plan_predicted_imbalance": round(max(plan_cost) / (sum(plan_cost) / len(plan_cost)), 4) if plan_cost else 1.0,
                      "measured_imbalance": round(max(rank_dts) / (sum(rank_dts) / len(rank_dts)), 4)},
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "one synthetic scene pair per step per GPU: 2 fragments x 5000 keypoints x 60 rotations x 32-D "
                                   "-> PartI group conv + invariant pooling -> mutual NN -> Des2R -> PartII -> YOHO-O (<=1000 hypotheses); "
                                   "random-init weights (seeded), inputs resident in HBM",
                       "ms_per_step_repeats": {"min": round(min(dts) / args.steps * 1e3, 3), "max": round(max(dts) / args.steps * 1e3, 3), "n": len(dts)},
                       "keypoints_per_fragment": KP, "partI_batch": nkp, "matches": M, "hypotheses": min(1000, M), "gconv": args.gconv, "partII": args.partII,
                       "pairs_per_step": pairs_per_step, "pairs_in_flight": args.in_flight, "pair_call": one_call,
                       "partI_schedule": ("breadth-first (every layer over the whole pass)" if sched_chunk == 0 else
                                          f"depth-first, chunks of {sched_chunk} keypoints on {sched_streams} stream(s)"),
                       "parallelism": (f"one pair per GPU per step, {world} GPU(s), no data-path collective" if args.scaling == "weak" else
                                       f"64 pairs (8 scenes x 8) per step dealt to {world} rank(s) by run_dataset.plan_shards, no data-path collective")},
            "roofline": roof,
            "roofline_extra": {"launch_ms": [round(float(v), 3) for v in conv_ms[:4]], "head_ms": round(float(conv_ms[4]), 3),
                               "tail_ms": round(float(conv_ms[5]), 3), "transform_ms": round(float(conv_ms[6]), 3), "hbm": hbm,
                               "pass_ms_one_stream": round(float(conv_ms[12]), 3),
                               "pass_ms_timed_schedule": round(pass_ms_timed_schedule, 3) if pass_ms_timed_schedule else round(float(conv_ms[12]), 3),
                               "power": {"timed_steps": power_steps, "profiled_partI_passes": power_prof,
                                         "note": "timed_steps: socket power read from the SMU (amdsmi; a ~1 s moving average) once, right behind the last timed "
                                                 "region, shader clock from the library's one-wave clock probe (shader cycles per constant-rate wall "
                                                 "tick, 20 us each, own high-priority stream) DURING the regions; profiled_partI_passes: the same two readings for the "
                                                 "profiled passes; nominal maximum 2400 MHz"},
                               "range_repeats": guard_total()},
            "range_guard": {"headline_repeats": headline_range_repeats, "all_legs_repeats": guard_total(),
                            "contexts": [c.range_report() for c in [ctx] + (streamer.desc + [streamer.est] if streamer else [])],
                            "note": "passes repeated in bf16x3 because a value left the fp16 planes' range (hip.Context._repeat_wider); 0 everywhere "
                                    "= every timed pass ran once in the fp16x2 arithmetic"},
        }
        if sustained is not None:
            out["sustained"] = sustained
        if yohoc is not None:
            out["yohoc"] = yohoc
        if sel_leg is not None:
            out["yohoo_selected_hypotheses"] = sel_leg
        if fgemm8 is not None:
            out["fgemm8"] = fgemm8
    # dataset-scale leg (BASELINE configs 3 / 5): 60 fragments x 5000 keypoints from .npy files on disk, ~500 pairs, through the
    # dataset driver; with N ranks the scene's pairs are dealt to them by run_dataset.plan_shards.  Never part of `value`.
    dataset = None
    if not args.no_dataset:
        streamer = None
        torch.cuda.empty_cache()
        try:
            import contextlib
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import bench_dataset
            with contextlib.redirect_stdout(sys.stderr):      # the evaluation code prints file names: stdout carries the one JSON line only
                dataset = bench_dataset.run(nfrag=60, kp=KP, span=9, estimator="yohoo", workdir=os.environ.get("YOHO_DS_WORKDIR", "/tmp/yoho_ds"), runs=2)
        except Exception as e:           # the headline must survive a failure of this leg
            dataset = {"error": f"{type(e).__name__}: {e}"}
    # raw-cloud leg: rank 0 only, the other ranks wait at the barrier below
    fcgf = None
    if not args.no_fcgf and rank == 0:
        streamer = None
        torch.cuda.empty_cache()
        try:
            fcgf = fcgf_leg(ctx, dev)
        except Exception as e:           # the headline must survive a failure of this leg
            fcgf = {"error": f"{type(e).__name__}: {e}"}
        out["fcgf"] = fcgf
    if rank == 0:
        if dataset is not None:
            out["dataset"] = dataset
        # last: the CPU legs (64+ host threads and a few GB of host arrays) stay out of the way of the device legs above
        if not args.no_cpu_baseline and world == 1:
            rb = pipeline.run_pair(ctx, f0, f1, k0, k1, max_iter=1000, order_rng=np.random.RandomState(7))       # this rank's first pair
            pr0 = {"feat0": f0.cpu().numpy(), "feat1": f1.cpu().numpy(), "keys0": k0.cpu().numpy(), "keys1": k1.cpu().numpy()}
            out["cpu_baseline"] = cpu_baseline(pr0, rb.eqv[0]["eqv"].cpu().numpy(), rb.eqv[1]["eqv"].cpu().numpy(), rb.match.cpu().numpy(),
                                               rb.dr_index.cpu().numpy())
            del rb, pr0
        print(json.dumps(out), flush=True)
    ydist.barrier()


if __name__ == "__main__":
    main()
