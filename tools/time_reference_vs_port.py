#!/usr/bin/env python
"""Build-container check behind bench.py's cpu_baseline: the REAL reference (imported from /root/reference with the
shims of oracle/gen_golden.py) against the oracle port that bench.py times on the GPU box, same synthetic pair, same
host cores, stage by stage - outputs equal to fixture tolerance, timings side by side.

    python tools/time_reference_vs_port.py [K]      ->  profiles/r02_reference_vs_port.json

The reference runs its own classes through its own .npy stage cache (that is its CPU path, disk included); the port
runs bench.py's cpu_baseline op sequence in memory.  /root/reference is read here only; nothing of it travels.
"""
import json
import os
import sys
import time
import types
import shutil
import tempfile
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)
import gen_golden as gg          # noqa: E402
import yoho_oracle as orc        # noqa: E402
from yoho_amd import weights as W, synth          # noqa: E402
from yoho_amd.tables import GroupTables           # noqa: E402


def main(K=300):
    torch.set_num_threads(os.cpu_count() or 1)
    tb = GroupTables(os.path.join(gg.REF, "group_related"))
    sd1 = W.synth_state_dict(W.PARTI_SPEC, 7)
    sd2 = W.synth_state_dict(W.PARTII_SPEC, 8)
    pr = synth.make_pair(K, seed=0, tables=tb)

    # ---- the port (bench.py cpu_baseline op sequence) ----
    tp = {}
    t0 = time.time()
    e0 = np.concatenate([orc.partI_forward_torch(pr["feat0"][s:s + 900], sd1, tb.N)[0] for s in range(0, K, 900)])
    e1 = np.concatenate([orc.partI_forward_torch(pr["feat1"][s:s + 900], sd1, tb.N)[0] for s in range(0, K, 900)])
    tp["partI"] = time.time() - t0
    t0 = time.time()
    m = orc.mutual_match(orc.group_mean_np(e0), orc.group_mean_np(e1))
    tp["matcher"] = time.time() - t0
    t0 = time.time()
    dr = orc.des2r_torch(e1[m[:, 1]], e0[m[:, 0]], tb.P)
    tp["des2r"] = time.time() - t0
    t0 = time.time()
    q = np.concatenate([orc.partII_forward_torch(pr["feat1"][m[s:s + 1000, 1]], pr["feat0"][m[s:s + 1000, 0]], e1[m[s:s + 1000, 1]],
                                                 e0[m[s:s + 1000, 0]], dr[s:s + 1000], sd2, tb.N, tb.P) for s in range(0, len(m), 1000)])
    k0, k1 = pr["keys0"][m[:, 0]], pr["keys1"][m[:, 1]]
    T = orc.hyp_from_quat(q, dr, k0, k1, tb.R32)
    tp["partII"] = time.time() - t0
    t0 = time.time()
    order = np.arange(len(m))
    np.random.seed(1234)
    np.random.shuffle(order)
    bid, cnt, Tb = orc.yohoo_select(k0, k1, T, order, 0.09, 1000)
    tp["yohoo"] = time.time() - t0

    # ---- the reference, through its own classes and stage cache ----
    sys.path.remove(REPO)
    network, extractor, matcher, estimator, knn_search, r_eval = gg.import_reference()
    work = tempfile.mkdtemp(prefix="yoho_time_")
    tr = {}
    try:
        model_fn = os.path.join(work, "model")
        for sub, sd in (("PartI_train", sd1), ("PartII_train", sd2)):
            os.makedirs(os.path.join(model_fn, sub))
            W.save_checkpoint(os.path.join(model_fn, sub, "model_best.pth"), sd, 0.5)

        def cfg(part):
            return types.SimpleNamespace(
                SO3_related_files=os.path.join(gg.REF, "group_related"), model_fn=model_fn, output_cache_fn=os.path.join(work, "cache"),
                origin_data_dir=os.path.join(work, "origin"), test_network_type=f"{part}_test", train_network_type=f"{part}_train",
                test_batch_size=900 if part == "PartI" else 1000, ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09)
        name = "synth/room"
        ds = gg.FakeDataset(name, ["0", "1"], [("0", "1")], {"0": pr["keys0"], "1": pr["keys1"]}, pr["gt"])
        cache = os.path.join(work, "cache", "Testset", name)
        os.makedirs(os.path.join(cache, "FCGF_Input_Group_feature"))
        np.save(os.path.join(cache, "FCGF_Input_Group_feature", "0.npy"), pr["feat0"])
        np.save(os.path.join(cache, "FCGF_Input_Group_feature", "1.npy"), pr["feat1"])
        kdir = os.path.join(work, "origin", name, "Keypoints_PC")
        os.makedirs(kdir)
        np.save(os.path.join(kdir, "cloud_bin_0Keypoints.npy"), pr["keys0"])
        np.save(os.path.join(kdir, "cloud_bin_1Keypoints.npy"), pr["keys1"])
        ex1 = extractor.extractor_PartI(cfg("PartI"))
        t0 = time.time(); ex1.Extract(ds); tr["partI"] = time.time() - t0
        mt = matcher.matcher_dual(cfg("PartI"))
        t0 = time.time(); mt.match(ds); tr["matcher"] = time.time() - t0
        dri = extractor.extractor_dr_index(cfg("PartI"))
        t0 = time.time(); dri.PartI_Rindex(ds); tr["des2r"] = time.time() - t0
        ex2 = extractor.extractor_PartII(cfg("PartII"))
        t0 = time.time(); ex2.PartII_R_pre(ds); tr["partII"] = time.time() - t0
        est = estimator.yohoo(cfg("PartII"))
        np.random.seed(1234)
        t0 = time.time(); est.ransac(ds, max_iter=1000); tr["yohoo"] = time.time() - t0
        r_e0 = np.load(os.path.join(cache, "YOHO_Output_Group_feature", "0.npy"))
        r_m = np.load(os.path.join(cache, "Match", "0-1.npy"))
        r_dr = np.load(os.path.join(cache, "Match", "DR_index", "0-1.npy"))
        r_T = np.load(os.path.join(cache, "Match", "Trans_pre", "0-1.npy"))
        r_z = np.load(os.path.join(cache, "Match", "YOHO_O", "1000iters", "0-1.npz"))
    finally:
        shutil.rmtree(work, ignore_errors=True)

    rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(np.max(np.abs(b)), 1e-30))
    same = {"eqv_rel_err": rel(e0, r_e0), "match_equal": bool(np.array_equal(m, r_m)), "dr_index_equal": bool(np.array_equal(dr, r_dr)),
            "trans_pre_rel_err": rel(T, r_T), "yohoo_recall_equal": bool(bid == int(r_z["recalltime"])),
            "yohoo_trans_rel_err": rel(Tb, r_z["trans"][:3])}
    tot_p, tot_r = sum(tp.values()), sum(tr.values())
    out = {"what": "real reference (its own classes + .npy stage cache) vs the oracle port bench.py times as cpu_baseline, same inputs, build container",
           "keypoints_per_fragment": K, "matches": int(len(m)), "cores": os.cpu_count(), "torch": torch.__version__, "numpy": np.__version__,
           "reference_s": {k: round(v, 3) for k, v in tr.items()}, "port_s": {k: round(v, 3) for k, v in tp.items()},
           "reference_total_s": round(tot_r, 3), "port_total_s": round(tot_p, 3),
           "reference_kp_per_s": round(2 * K / tot_r, 1), "port_kp_per_s": round(2 * K / tot_p, 1),
           "port_over_reference_time": round(tot_p / tot_r, 3), "outputs": same}
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    with open(os.path.join(REPO, "profiles", "r02_reference_vs_port.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
