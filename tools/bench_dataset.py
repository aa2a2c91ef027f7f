"""Dataset-scale measurement (BASELINE config 3 / 5 shape): a synthetic 3DMatch-shaped scene - 60 fragments x 5000 keypoints, the
FCGF group features as 38.4 MB .npy files on disk in the reference's cache layout, ~500 scene pairs - through
run_dataset.eval_sharded (load -> H2D -> PartI once per fragment -> HBM-resident pairs -> pre.log -> Registration Recall).

    python tools/bench_dataset.py [--nfrag 60] [--kp 5000] [--span 9] [--estimator yohoo|yohoc] [--workdir /tmp/yoho_ds] [--runs 2]
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_dataset.py        (the scene's pairs are dealt to the N ranks)

The reference pays ~300 MB of disk reads per PAIR (tests/extractor.py:166-169, tests/matcher.py:33-36: four 38.4 MB feature files
+ two descriptor files); here a fragment is read once per rank.  Prints one JSON object; bench.py embeds the same dict as "dataset".
"""
import argparse
import json
import os
import shutil
import sys
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def build_scene(root, cache_scene_dir, nfrag, kp, span, seed=0, device=None, npairs=None):
    """A scene in the shape of a 3DMatch test scene: every fragment is a moved, row-shuffled, group-permuted, noisy copy of one base
    fragment (30 % of the rows replaced by outliers); pairs (i, j) with 0 < j - i <= span.  Generated with torch on `device` (the
    GPU when there is one: a throughput workload, not a fixture).  Writes gt.log / gt.info / keypoints / FCGF_Input_Group_feature
    files; returns the pair list."""
    import torch
    from yoho_amd.tables import default_tables
    tb = default_tables()
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    rs = np.random.RandomState(seed)
    unit = lambda t: t / torch.linalg.vector_norm(t, dim=1, keepdim=True)
    base = unit(torch.randn((kp, 32, 60), generator=gen, device=device, dtype=torch.float32))
    base_k = rs.rand(kp, 3) * 3.0
    P = torch.from_numpy(np.asarray(tb.P, dtype=np.int64)).to(device)
    os.makedirs(f"{root}/PointCloud", exist_ok=True)
    os.makedirs(f"{root}/Keypoints_PC", exist_ok=True)
    os.makedirs(f"{cache_scene_dir}/FCGF_Input_Group_feature", exist_ok=True)
    poses = []
    for f in range(nfrag):
        gi = int(rs.randint(60))
        ax = rs.randn(3); ax /= np.linalg.norm(ax)
        ang = np.deg2rad(rs.rand() * 4.0)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        Rres = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        R = Rres @ tb.R64[gi]
        t = (rs.rand(3) - 0.5) * 2.0
        perm = rs.permutation(kp)
        out = rs.rand(kp) < 0.3
        ff = base[torch.from_numpy(perm).to(device)][:, :, P[gi]] + 0.02 * torch.randn((kp, 32, 60), generator=gen, device=device, dtype=torch.float32)
        od = torch.from_numpy(out).to(device)
        ff[od] = torch.randn((int(out.sum()), 32, 60), generator=gen, device=device, dtype=torch.float32)
        ff = unit(ff).contiguous().cpu().numpy()
        kk = base_k[perm] @ R.T + t + 0.01 * rs.standard_normal((kp, 3))
        kk[out] = rs.rand(int(out.sum()), 3) * 3.0
        np.save(f"{cache_scene_dir}/FCGF_Input_Group_feature/{f}.npy", ff)
        np.save(f"{root}/Keypoints_PC/cloud_bin_{f}Keypoints.npy", np.ascontiguousarray(kk))
        poses.append((R, t))
    pairs = [(i, j) for i in range(nfrag) for j in range(i + 1, min(nfrag, i + span + 1))]
    if npairs is not None:
        # exactly npairs pairs, nearest fragments first (a gt.log lists the overlapping pairs, which are mostly close in scan order)
        allp = sorted(((i, j) for i in range(nfrag) for j in range(i + 1, nfrag)), key=lambda p: (p[1] - p[0], p[0]))
        pairs = sorted(allp[:npairs])
    with open(f"{root}/PointCloud/gt.log", "w") as fl, open(f"{root}/PointCloud/gt.info", "w") as fi:
        for (i, j) in pairs:                       # keys_i = Ri Rj^T (keys_j - tj) + ti
            (Ri, ti), (Rj, tj) = poses[i], poses[j]
            T = np.eye(4)
            T[:3, :3] = Ri @ Rj.T
            T[:3, 3] = ti - T[:3, :3] @ tj
            fl.write(f"{i}\t{j}\t{nfrag}\n" + "".join("\t".join(repr(float(v)) for v in T[r]) + "\n" for r in range(4)))
            fi.write(f"{i}\t{j}\t{nfrag}\n" + "".join("\t".join(repr(float(1.0 if r == c else 0.0)) for c in range(6)) + "\n" for r in range(6)))
    return pairs


def drop_page_cache(root):
    """evict the files under `root` from the page cache (posix_fadvise DONTNEED per file - this process's own files only, no machine-wide
    setting is touched); True when every file was advised"""
    try:
        os.sync()
        n = 0
        for d, _, files in os.walk(root):
            for fn in files:
                fd = os.open(os.path.join(d, fn), os.O_RDONLY)
                try:
                    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                    n += 1
                finally:
                    os.close(fd)
        return n > 0
    except Exception:
        return False


# the 3DMatch test set's shape: fragments per scene (utils/dataset.py:163-167) and pairs per scene as its gt.log files list them (1623)
PRESET_3DMATCH = [("kitchen", 60, 506), ("home1", 60, 156), ("home2", 60, 208), ("hotel1", 55, 226), ("hotel2", 57, 104), ("hotel3", 37, 54),
                  ("study", 66, 292), ("lab", 38, 77)]


def run(nfrag=60, kp=5000, span=9, estimator="yohoo", workdir="/tmp/yoho_ds", runs=2, max_iter=1000, keep=False, preset=None, hypotheses="selected", pair_workers=2, fused=True, overlap=True):
    """-> dict for the bench line.  Every rank of an initialised process group calls this; rank 0 builds the files.
    preset="3dmatch": eight scenes with the fragment and pair counts of the 3DMatch test set instead of one scene."""
    import torch
    from yoho_amd import hip, weights as W, run_dataset, dist as ydist
    from yoho_amd.dataset import ThrDMatchPartDataset
    rank, world, local = ydist.init_from_env()
    scenes = [("scene0", nfrag, None)] if preset is None else PRESET_3DMATCH
    cache = f"{workdir}/cache"
    def all_ok(ok, what):
        """every rank learns whether any rank failed (a failure on one rank must not leave the others waiting at a barrier)"""
        if ydist.max_over_ranks(0.0 if ok else 1.0) > 0.0:
            raise RuntimeError(f"{what} failed on at least one rank" + ("" if ok else f" (this one, rank {rank}): {err[0]}"))
    err = [None]
    t0 = time.perf_counter()
    ok = True
    if rank == 0:
        try:
            shutil.rmtree(workdir, ignore_errors=True)
            for si, (sn, nf, npairs) in enumerate(scenes):
                build_scene(f"{workdir}/origin/synthds/{sn}", f"{cache}/Testset/synthds/{sn}", nf, kp, span, seed=si, npairs=npairs)
        except Exception as e:
            ok, err[0] = False, f"{type(e).__name__}: {e}"
    all_ok(ok, "building the scene files")
    t_build = time.perf_counter() - t0
    datasets = {"wholesetname": "synthds"}
    for sn, nf, _ in scenes:
        ds = ThrDMatchPartDataset(f"{workdir}/origin/synthds/{sn}", nf)
        ds.name = f"synthds/{sn}"
        datasets[sn] = ds
    nfrag_all = sum(nf for _, nf, _ in scenes)
    npairs_all = sum(len(datasets[sn].pair_ids) for sn, _, _ in scenes)
    cfg = types.SimpleNamespace(SO3_related_files=None, model_fn=f"{workdir}/model", output_cache_fn=cache, origin_data_dir=f"{workdir}/origin",
                                ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09, RR_dist_threshold=0.2, testset_name="synthds")
    sd1 = W.synth_state_dict(W.PARTI_SPEC, 7)
    sd2 = W.identity_head(W.synth_state_dict(W.PARTII_SPEC, 8))
    ctx = hip.Context(local if world > 1 else torch.cuda.current_device())
    t0 = time.perf_counter()
    ctx.load_partI(sd1)                            # once per process (host-side weight packing: BN folding, irrep-GEMM planes)
    if estimator == "yohoo":
        ctx.load_partII(sd2)
    t_weights = time.perf_counter() - t0
    out = {"workload": (f"synthetic scene, {nfrag} fragments x {kp} keypoints" if preset is None else
                        f"synthetic test set in the shape of 3DMatch's: 8 scenes, {nfrag_all} fragments x {kp} keypoints") +
                       f" ({nfrag_all * kp * 7680 / 1e9:.2f} GB of FCGF group features as .npy on disk), {npairs_all} pairs, estimator {estimator}, {world} rank(s)",
           "fragments": nfrag_all, "keypoints_per_fragment": kp, "pairs": npairs_all, "ranks": world, "estimator": estimator,
           "hypotheses": ("PartII only for the <= %d matches the YOHO-O vote reads" % max_iter) if hypotheses == "selected" else "PartII for every match (the reference's Trans_pre stage)",
           "scenes": {sn: {"fragments": nf, "pairs": len(datasets[sn].pair_ids)} for sn, nf, _ in scenes},
           "scene_build_s": round(t_build, 2), "load_weights_once_s": round(t_weights, 3), "runs": []}
    for r in range(runs):
        cold = drop_page_cache(workdir) if (r == 0 and rank == 0) else False
        ydist.barrier()
        torch.cuda.synchronize()
        stats = {}
        t0 = time.perf_counter()
        ok, rr = True, None
        try:
            rr = run_dataset.eval_sharded(cfg, max_iter=max_iter, estimator=estimator, datasets=datasets, base_seed=r, ctx=ctx, state_dicts=(sd1, sd2), weights_loaded=True, hypotheses=hypotheses, pair_workers=pair_workers, fused=fused, overlap=overlap,
                                          stats_out=stats)
            torch.cuda.synchronize()
        except Exception as e:
            ok, err[0] = False, f"{type(e).__name__}: {e}"
        all_ok(ok, "eval_sharded")
        dt = ydist.max_over_ranks(time.perf_counter() - t0)
        res = stats.pop("results")
        if rank == 0:
            import hashlib
            out["trans_sha256"] = hashlib.sha256(b"".join(np.ascontiguousarray(p["trans"], dtype=np.float64).tobytes() + int(p["recalltime"]).to_bytes(8, "little", signed=True)
                                                          for sn, _, _ in scenes for p in res[sn])).hexdigest()      # of the last run: every pair's transform + recalltime
            inl = [p["inliers"] for sn, _, _ in scenes for p in res[sn]]
            mt = [p["matches"] for sn, _, _ in scenes for p in res[sn]]
            pw = stats.get("pairs_wall_s", stats["pairs_s"])            # from the first pair worker's start to the last pair's end (overlaps the setup)
            tail = stats.get("pairs_tail_s", pw)                       # what of it is left behind the last fragment's description
            wall = stats.get("parts_wall_s", stats["setup_s"] + pw)
            row = {"page_cache": "dropped before the run" if cold else "warm", "total_s": round(dt, 3),
                   "keypoints_per_s_end_to_end": round(nfrag_all * kp / dt, 1), "pairs_per_s_end_to_end": round(npairs_all / dt, 1),
                   "rank0": {"fragments": stats["fragments"], "pairs": stats["pairs"],
                             "setup_s (load + H2D + PartI, overlapped)": round(stats["setup_s"], 3),
                             "disk_read_s (loader thread)": round(stats["load_s"], 3), "disk_GBps": round(stats["bytes_read"] / max(stats["load_s"], 1e-9) / 1e9, 2),
                             "device_waiting_for_loader_s": round(stats["load_wait_s"], 3), "h2d_plus_partI_s": round(stats["h2d_describe_s"], 3),
                             "fragments_per_s (load + describe)": round(stats["fragments"] / max(stats["setup_s"], 1e-9), 1),
                             "setup_and_pairs_wall_s": round(wall, 3), "pairs_behind_last_setup_s": round(tail, 3),
                             "pairs_per_s": round(stats["pairs"] / max(wall, 1e-9), 1),
                             "ms_per_pair (wall of setup + pairs / pairs)": round(wall / max(stats["pairs"], 1) * 1e3, 3),
                             "pair_workers": pair_workers, "one_call_per_pair": bool(fused), "pairs_overlap_setup": bool(overlap),
                             "gather_write_RR_s": round(dt - wall, 3)},
                   "registration_recall": rr, "mean_matches": round(float(np.mean(mt)), 1), "mean_inliers_of_winner": round(float(np.mean(inl)), 1)}
            out["runs"].append(row)
    if rank == 0 and not keep:
        shutil.rmtree(workdir, ignore_errors=True)
    ydist.barrier()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nfrag", type=int, default=60)
    ap.add_argument("--kp", type=int, default=5000)
    ap.add_argument("--span", type=int, default=9)
    ap.add_argument("--estimator", default="yohoo")
    ap.add_argument("--workdir", default="/tmp/yoho_ds")
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--preset", default=None, choices=[None, "3dmatch"])
    ap.add_argument("--hypotheses", default="selected", choices=["selected", "all"])
    ap.add_argument("--pair-workers", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true", help="part by part: load + describe all fragments of a scene part, then run its pairs")
    ap.add_argument("--staged", action="store_true", help="compose every pair from the staged entries in Python (pipeline.run_pair) instead of yoho_register_pair")
    a = ap.parse_args()
    o = run(a.nfrag, a.kp, a.span, a.estimator, a.workdir, a.runs, preset=a.preset, hypotheses=a.hypotheses, pair_workers=a.pair_workers, fused=not a.staged, overlap=not a.no_overlap)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(o), flush=True)
