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
Besides the headline (descriptor + YOHO-O) the line carries "yohoc": the same step with the YOHO-C estimator (config 5:
descriptor + 1000 RANSAC iterations sampled on the device, no PartII).  Prints ONE JSON line on rank 0.
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


PMC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")     # newest first


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


def cpu_baseline(K=600):
    """The oracle (a port of the reference's op sequence, torch-CPU kernels for the convs exactly as
    the reference's CPU path, numpy for the rest) on a bounded sample of the same workload:
    one synthetic pair with K keypoints per fragment, test_batch_size 900 / 1000."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import yoho_oracle as orc
    from yoho_amd.tables import default_tables
    tb = default_tables()
    torch.set_num_threads(os.cpu_count() or 1)
    sd1 = W.synth_state_dict(W.PARTI_SPEC, 7)
    sd2 = W.synth_state_dict(W.PARTII_SPEC, 8)
    pr = synth.make_pair(K, seed=0)
    t0 = time.time()
    e0 = np.concatenate([orc.partI_forward_torch(pr["feat0"][s:s + 900], sd1, tb.N)[0] for s in range(0, K, 900)])
    e1 = np.concatenate([orc.partI_forward_torch(pr["feat1"][s:s + 900], sd1, tb.N)[0] for s in range(0, K, 900)])
    t_desc = time.time() - t0
    m = orc.mutual_match(orc.group_mean_np(e0), orc.group_mean_np(e1))
    dr = orc.des2r_torch(e1[m[:, 1]], e0[m[:, 0]], tb.P)
    q = np.concatenate([orc.partII_forward_torch(pr["feat1"][m[s:s + 1000, 1]], pr["feat0"][m[s:s + 1000, 0]], e1[m[s:s + 1000, 1]],
                                                 e0[m[s:s + 1000, 0]], dr[s:s + 1000], sd2, tb.N, tb.P) for s in range(0, len(m), 1000)])
    k0, k1 = pr["keys0"][m[:, 0]], pr["keys1"][m[:, 1]]
    T = orc.hyp_from_quat(q, dr, k0, k1, tb.R32)
    order = np.arange(len(m))
    np.random.RandomState(0).shuffle(order)
    orc.yohoo_select(k0, k1, T, order, 0.09, 1000)
    dt = time.time() - t0
    out = {"value": round(2 * K / dt, 2), "unit": "keypoints/s", "cores": os.cpu_count(), "kind": "port",
           "sample": f"one synthetic pair, {K} keypoints/fragment ({len(m)} matches): PartI (torch-CPU conv2d, bs=900) "
                     f"{t_desc:.1f}s of {dt:.1f}s total, then matcher (numpy) + Des2R + PartII (torch-CPU ops of the reference, bs=1000) "
                     f"+ YOHO-O (numpy)"}
    chk = os.path.join(REPO, "profiles", "r02_reference_vs_port.json")
    if os.path.exists(chk):
        # the port against the real reference in the build container, same inputs (tools/time_reference_vs_port.py)
        c = json.load(open(chk))
        out["reference_timing_check"] = {k: c[k] for k in ("keypoints_per_fragment", "cores", "reference_total_s", "port_total_s",
                                                            "port_over_reference_time", "outputs")}
        out["reference_timing_check"]["source"] = "profiles/r02_reference_vs_port.json (tools/time_reference_vs_port.py, build container)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-yohoc", action="store_true", help="skip the YOHO-C leg")
    ap.add_argument("--in-flight", type=int, choices=[1, 2], default=2,
                    help="pairs in flight: 2 (default) queues the descriptor pass of the next pair on a second HIP stream before waiting for "
                         "the current pair's read-backs (pipeline.PairStreamer); 1 runs the pairs strictly one after the other")
    ap.add_argument("--gconv", choices=["f32", "bf16x3", "fourier", "fp16x2", "fgemm", "fgemm256", "fgemm128"], default=os.environ.get("YOHO_GCONV", "fgemm"),
                    help="PartI group conv: group-Fourier domain (fp32 MFMA), direct fp32 MFMA, direct 3-way bf16 split MFMA, "
                         "or direct 2-way fp16 split MFMA")
    ap.add_argument("--partII", choices=["f32", "bf16x3", "fp16x2"], default=os.environ.get("YOHO_PARTII", "fp16x2"),
                    help="arithmetic of the two PartII cone layers")
    ap.add_argument("--repeats", type=int, default=3, help="the timed region (exactly --steps steps) is run this many times; ms_per_step / value are "
                                                           "the median repeat, min / max are reported beside it")
    ap.add_argument("--partI-schedule", default=os.environ.get("YOHO_PARTI_CHUNK", DEFAULT_SCHEDULE),
                    help="PartI pass: '0' breadth-first, 'C' or 'CxS' depth-first over chunks of C keypoints on S (1 or 2) streams")
    ap.add_argument("--no-dataset", action="store_true", help="skip the dataset-scale leg (tools/bench_dataset.py: 60 fragments from disk, ~500 pairs)")
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
    else:
        # a fixed list of 64 pairs = 8 scenes x 8 pairs, dealt to the ranks by the dataset driver's plan (scenes whole, a
        # scene larger than a rank's share cut round-robin).  8 distinct synthetic pairs, each listed 8 times.
        from yoho_amd.run_dataset import plan_shards
        plan = plan_shards({f"scene{i}": 8 for i in range(8)}, world)[rank]
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
            return streamer.run(todo, inlier_dist=dist, max_iter=1000, order_rng=rng, estimator=estimator,
                                seeds=[seed0 + i for i in range(len(todo))], hypotheses=hypotheses)[-1]
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

    dt, dts, rank_dts, res, power_steps = timed("yohoo", args.steps, max(args.warmup, 1), args.repeats)
    # the same step with PartII evaluated only for the 1000 matches the YOHO-O vote reads (pipeline.run_pair hypotheses="selected":
    # identical winner / transform, tests/test_gpu_fullsize.py); reported beside the headline, which keeps the reference's
    # workload (PartII and [R|t] for every match, as its Trans_pre stage leaves them)
    sel_leg = None
    if not args.no_yohoc:
        dts_, _, _, ress, _ = timed("yohoo", args.steps, max(min(args.warmup, 2), 1), 1, "selected")
        sel_leg = {"metric": "keypoints/sec (5000 kp x60 rot desc + YOHO-O, PartII only for the 1000 voted hypotheses)",
                   "value": round(pairs_per_step * 2 * KP * args.steps / dts_, 1), "ms_per_step": round(dts_ / args.steps * 1e3, 3),
                   "winner_inliers": int(ress.best_count)}
        # one pair both ways with the same shuffle: the winner and the transform must be the same
        ra_ = pipeline.run_pair(ctx, f0, f1, k0, k1, max_iter=1000, order_rng=np.random.RandomState(7))
        rs_ = pipeline.run_pair(ctx, f0, f1, k0, k1, max_iter=1000, order_rng=np.random.RandomState(7), eqv=ra_.eqv, hypotheses="selected")
        sel_leg["same_winner_and_transform_as_all"] = bool((ra_.best_h, ra_.best_count) == (rs_.best_h, rs_.best_count) and
                                                           np.array_equal(np.asarray(ra_.trans), np.asarray(rs_.trans)))
    yohoc = None
    if not args.no_yohoc:
        dtc, _, _, resc, _ = timed("yohoc", args.steps, max(min(args.warmup, 2), 1))
        # host time per pair of the estimator call alone (launches only: nothing is read back inside the call)
        m_, dr_ = res.match, res.dr_index
        torch.cuda.synchronize()
        th = time.perf_counter()
        for i in range(20):
            ctx.c_ransac_device(k0, k1, dr_, 1000, 100 + i, 0.07, match=m_)
        host_ms = (time.perf_counter() - th) / 20 * 1e3
        torch.cuda.synchronize()
        yohoc = {"metric": "keypoints/sec (5000 kp x60 rot desc + YOHO-C, 1000 iterations sampled on the device)",
                 "value": round(pairs_per_step * 2 * KP * args.steps / dtc, 1), "ms_per_step": round(dtc / args.steps * 1e3, 3),
                 "iterations": 1000, "estimator_host_ms_per_pair": round(host_ms, 4),
                 "winner_inliers": int(resc.best_count), "matches": int(resc.match.shape[0])}

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
    if args.gconv in ("fgemm", "fgemm256", "fgemm128"):
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

    if rank == 0:
        M = int(res.match.shape[0])
        if args.gconv == "f32":
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK, 4), "traffic": pmc_traffic("f32")[0],
                    "kernel": "gconv_kernel<15,false> (4 launches = 4 PartI layers, 2.1725 algorithmic TFLOP per 5000 kp)"}
            dtype = "f32"
        elif args.gconv == "bf16x3":
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / BF16_MFMA_PEAK, 4), "traffic": pmc_traffic("bf16x3")[0],
                    "kernel": "gconv16_kernel<15,2> + <8,1> (4 launches = 4 PartI layers, 2.1725 algorithmic TFLOP per 5000 kp)",
                    "executed_tflops": round(achieved * BF16X3_EXEC_PER_ALG, 1),
                    "executed_frac": round(achieved * BF16X3_EXEC_PER_ALG / BF16_MFMA_PEAK, 4),
                    "note": "achieved = algorithmic fp32-equivalent FLOP/s; the fp32-accurate bf16 split issues 6.46 bf16 MFMA flops "
                            "per algorithmic flop, so frac <= 0.155 for this formulation"}
            dtype = "bf16x3 split (fp32-accurate, fp32 accumulate)"
        elif args.gconv == "fp16x2":
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / BF16_MFMA_PEAK, 4), "traffic": pmc_traffic("fp16x2")[0],
                    "kernel": "gconv16_kernel<15,2,2> + <8,1,2> (4 launches = 4 PartI layers, 2.1725 algorithmic TFLOP per 5000 kp)",
                    "executed_tflops": round(achieved * BF16X3_EXEC_PER_ALG / 2, 1),
                    "executed_frac": round(achieved * BF16X3_EXEC_PER_ALG / 2 / BF16_MFMA_PEAK, 4),
                    "note": "achieved = algorithmic fp32-equivalent FLOP/s; the 2-way fp16 split (x = hi + lo, 3 products, error "
                            "<= 3*2^-22 per product) issues 3.23 fp16 MFMA flops per algorithmic flop, so frac <= 0.31"}
            dtype = "fp16x2 split (2^-22-accurate products, fp32 accumulate)"
        elif args.gconv in ("fgemm", "fgemm256", "fgemm128"):
            issued = fgemm_issued_flops(nkp, args.gconv) / (gconv_total_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP16_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / FP16_MFMA_PEAK, 4), "traffic": pmc_traffic(args.gconv)[0], "traffic_source": pmc_traffic(args.gconv)[1],
                    "kernel": {"fgemm": "fgemm3_kernel (fgemm3s_kernel for the 32-channel layer)", "fgemm128": "fgemm2_kernel", "fgemm256": "fgemm_kernel"}[args.gconv] +
                              " (4 launches = 4 PartI layers over both fragments, 4.345 algorithmic TFLOP per 10000 kp)",
                    "executed_tflops": round(issued, 1), "executed_frac": round(issued / FP16_MFMA_PEAK, 4),
                    "executed_frac_per_launch": [round(f / (ms * 1e-3) / 1e12 / FP16_MFMA_PEAK, 4)
                                                 for f, ms in zip(fgemm_issued_flops_per_layer(nkp, args.gconv), conv_ms[:4])],
                    "conv_total_frac": round(FLOP_PER_KP * nkp / (conv_total_ms * 1e-3) / 1e12 / FP16_MFMA_PEAK, 4),
                    "conv_total_ms": round(conv_total_ms, 3),
                    "note": "achieved = algorithmic FLOP/s of the reference's direct 13-tap formulation (SURVEY 8d) over the 4 "
                            "fgemm launches. The kernel evaluates the same convolution on group-Fourier coefficients as five dense "
                            "irrep GEMMs (244/780 of the multiply-adds) with every product as 3 fp16 MFMA products (fp16x2 split, "
                            "fp32 accumulate); executed_tflops / executed_frac = fp16 MFMA flops actually issued (padding "
                            "included) against the dense fp16 peak. The transform kernels between the layers are timed "
                            "separately (roofline_extra.transform_ms)"}
            dtype = "fp16x2 split (2^-22-accurate products, fp32 accumulate)"
        else:
            ex = achieved * FOURIER_EXEC_PER_ALG
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK, 4), "traffic": pmc_traffic("fourier")[0],
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
            "ranks": {"world_size_seen": world, "backend": (torch.distributed.get_backend() if world > 1 else None),
                      "ms_per_step_per_rank": {"min": round(min(rank_dts) / args.steps * 1e3, 3), "mean": round(float(np.mean(rank_dts)) / args.steps * 1e3, 3),
                                               "max": round(max(rank_dts) / args.steps * 1e3, 3), "all": [round(v / args.steps * 1e3, 3) for v in rank_dts]}},
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "one synthetic scene pair per step per GPU: 2 fragments x 5000 keypoints x 60 rotations x 32-D "
                                   "-> PartI group conv + invariant pooling -> mutual NN -> Des2R -> PartII -> YOHO-O (<=1000 hypotheses); "
                                   "random-init weights (seeded), inputs resident in HBM",
                       "keypoints_per_fragment": KP, "partI_batch": nkp, "matches": M, "hypotheses": min(1000, M), "gconv": args.gconv, "partII": args.partII,
                       "pairs_per_step": pairs_per_step, "pairs_in_flight": args.in_flight,
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
                               "range_repeats": int(ctx.range_fallbacks + (sum(c.range_fallbacks for c in streamer.desc + [streamer.est]) if streamer else 0))},
        }
        if yohoc is not None:
            out["yohoc"] = yohoc
        if sel_leg is not None:
            out["yohoo_selected_hypotheses"] = sel_leg
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
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
    if rank == 0:
        if dataset is not None:
            out["dataset"] = dataset
        print(json.dumps(out), flush=True)
    ydist.barrier()


if __name__ == "__main__":
    main()
