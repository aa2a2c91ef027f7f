"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Runs only in the build container (the reference is pure Python on this path; it is not
copied, and this script is never imported by tests or the product).  The reference needs a
few import shims on a modern stack:
  np.int / np.float aliases, .cuda() -> identity (no GPU here), a stub `tensorboardX`,
  argv reset (argparse at import, utils/network.py:8).

Inputs are produced by yoho_amd.weights.hash_uniform (counter hash, reproducible anywhere), so
the fixtures hold seeds + the reference's outputs (+ small inputs where convenient).

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
import os
import sys
import types
import shutil
import tempfile
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from yoho_amd import weights as W          # noqa: E402  (generator + spec only; no HIP)
from yoho_amd.tables import GroupTables    # noqa: E402
from yoho_amd.synth import make_pair, unit_features, make_scene, write_scene_files  # noqa: E402


def import_reference():
    np.int = int
    np.float = float
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.modules["tensorboardX"] = types.SimpleNamespace(SummaryWriter=object)
    sys.modules["open3d"] = types.ModuleType("open3d")          # utils/dataset.py imports it; only .ply reading would use it
    sys.argv = ["x"]
    sys.path.insert(0, REF)
    import utils.network as network
    import tests.extractor as extractor      # the reference's tests/ package (pipeline plug-ins)
    import tests.matcher as matcher
    import tests.estimator as estimator
    import utils.knn_search as knn_search
    import utils.r_eval as r_eval
    # nibabel is not installed: utils/RR_cal.py only uses nibabel.quaternions.mat2quat, which is the same
    # K-matrix + eigh algorithm as the reference's own utils/r_eval.quaternion_from_matrix - use that.
    nq = types.ModuleType("nibabel.quaternions")
    nq.mat2quat = lambda M: r_eval.quaternion_from_matrix(np.asarray(M, dtype=np.float64))
    nib = types.ModuleType("nibabel")
    nib.quaternions = nq
    sys.modules["nibabel"] = nib
    sys.modules["nibabel.quaternions"] = nq
    return network, extractor, matcher, estimator, knn_search, r_eval


class FakeDataset:
    """Duck type of utils/dataset.py:ThrDMatchPartDataset as used by the path."""
    def __init__(self, name, pc_ids, pair_ids, kps, gt):
        self.name, self.pc_ids, self.pair_ids, self._kps, self._gt = name, pc_ids, pair_ids, kps, gt

    def get_kps(self, pc_id):
        return self._kps[pc_id]

    def get_transform(self, id0, id1):
        return self._gt


def main():
    # the reference has its own top-level `tests` package (the pipeline plug-ins): make sure the
    # repo root (which also has a tests/ directory) is not ahead of it on sys.path.
    sys.path.remove(REPO)
    network, extractor, matcher, estimator, knn_search, r_eval = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    tb = GroupTables(os.path.join(REF, "group_related"))
    so3 = os.path.join(REF, "group_related")
    seed = 7
    sd1 = W.synth_state_dict(W.PARTI_SPEC, seed)
    sd2 = W.synth_state_dict(W.PARTII_SPEC, seed + 1)

    work = tempfile.mkdtemp(prefix="yoho_gold_")
    try:
        model_fn = os.path.join(work, "model")
        for sub, sd in (("PartI_train", sd1), ("PartII_train", sd2)):
            os.makedirs(os.path.join(model_fn, sub))
            W.save_checkpoint(os.path.join(model_fn, sub, "model_best.pth"), sd, 0.5)

        def cfg(part):
            return types.SimpleNamespace(
                SO3_related_files=so3, model_fn=model_fn, output_cache_fn=os.path.join(work, "cache"),
                origin_data_dir=os.path.join(work, "origin"),
                test_network_type=f"{part}_test", train_network_type=f"{part}_train",
                test_batch_size=40 if part == "PartI" else 50,   # forces several chunks at K=96
                ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09)

        # ---------------- 1. PartI forward -------------------------------------------------
        net1 = network.PartI_test(cfg("PartI"))
        net1.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd1.items()})
        net1.eval()
        x = unit_features(16, seed=11)
        with torch.no_grad():
            o = net1(torch.from_numpy(x))
        np.savez(os.path.join(GOLD, "partI.npz"), wseed=seed, xseed=11, x=x,
                 eqv=o["eqv"].numpy(), inv=o["inv"].numpy())

        # equivariance property on the reference itself (documented in SURVEY section 4)
        with torch.no_grad():
            oi = net1(torch.from_numpy(np.ascontiguousarray(x[:, :, tb.P[17]])))
        assert np.array_equal(oi["eqv"].numpy(), o["eqv"].numpy()[:, :, tb.P[17]])

        # ---------------- 2. torch CPU reduction order used by pdist -----------------------
        rs = np.random.RandomState(5)
        A = rs.randn(120, 32).astype(np.float32) * 0.2
        B = rs.randn(250, 32).astype(np.float32) * 0.2
        knn = knn_search.knn_module.KNN(1)
        dist = knn.pdist(torch.from_numpy(A), torch.from_numpy(B), "L2").numpy()
        d2 = knn.pdist(torch.from_numpy(A), torch.from_numpy(B), "SquareL2").numpy()
        np.savez(os.path.join(GOLD, "pdist.npz"), A=A, B=B, dist=dist, d2=d2)

        # ---------------- 3..9 full chain on a synthetic pair ------------------------------
        K = 96
        pair = make_pair(K, seed=3, tables=tb, noise=0.02, outlier_frac=0.4)
        name = "synth/room"
        ds = FakeDataset(name, ["0", "1"], [("0", "1")], {"0": pair["keys0"], "1": pair["keys1"]},
                         pair["gt"])
        cache = os.path.join(work, "cache", "Testset", name)
        os.makedirs(os.path.join(cache, "FCGF_Input_Group_feature"))
        np.save(os.path.join(cache, "FCGF_Input_Group_feature", "0.npy"), pair["feat0"])
        np.save(os.path.join(cache, "FCGF_Input_Group_feature", "1.npy"), pair["feat1"])
        kdir = os.path.join(work, "origin", name, "Keypoints_PC")
        os.makedirs(kdir)
        np.save(os.path.join(kdir, "cloud_bin_0Keypoints.npy"), pair["keys0"])
        np.save(os.path.join(kdir, "cloud_bin_1Keypoints.npy"), pair["keys1"])

        extractor.extractor_PartI(cfg("PartI")).Extract(ds)
        eqv0 = np.load(os.path.join(cache, "YOHO_Output_Group_feature", "0.npy"))
        eqv1 = np.load(os.path.join(cache, "YOHO_Output_Group_feature", "1.npy"))
        matcher.matcher_dual(cfg("PartI")).match(ds)
        pps = np.load(os.path.join(cache, "Match", "0-1.npy"))
        dri = extractor.extractor_dr_index(cfg("PartI"))
        dri.PartI_Rindex(ds)
        dr = np.load(os.path.join(cache, "Match", "DR_index", "0-1.npy"))
        with torch.no_grad():
            d1 = torch.from_numpy(eqv1[pps[:, 1]]); d0 = torch.from_numpy(eqv0[pps[:, 0]])
            Bn, Fn, Gn = d1.shape
            cor = torch.einsum("bfag,bfg->ba", d1[:, :, dri.Nei_in_SO3].reshape([Bn, Fn, 60, 60]), d0).numpy()
        ex2 = extractor.extractor_PartII(cfg("PartII"))
        ex2.PartII_R_pre(ds)
        trans_pre = np.load(os.path.join(cache, "Match", "Trans_pre", "0-1.npy"))

        # PartII forward in isolation on the matched rows (first 16)
        f0, f1 = pair["feat0"][pps[:, 0]], pair["feat1"][pps[:, 1]]
        y0, y1 = eqv0[pps[:, 0]], eqv1[pps[:, 1]]
        batch = ex2.batch_create(f0, f1, y0, y1, dr, 0, 16)
        net2 = ex2.network
        net2.eval()
        with torch.no_grad():
            q = net2({k: v.clone() for k, v in batch.items()})["quaternion_pre"].numpy()

        np.random.seed(1234)
        est_o = estimator.yohoo(cfg("PartII"))
        est_o.ransac(ds, max_iter=1000)
        zo = np.load(os.path.join(cache, "Match", "YOHO_O", "1000iters", "0-1.npz"))
        prelog_o = open(os.path.join(cache, "Match", "YOHO_O", "1000iters", "pre.log")).read()
        # few-iteration run so that max_iter < M is exercised too
        np.random.seed(4321)
        est_o.ransac(ds, max_iter=20)
        zo20 = np.load(os.path.join(cache, "Match", "YOHO_O", "20iters", "0-1.npz"))

        np.random.seed(99)
        est_c = estimator.yohoc(cfg("PartI"))
        est_c.ransac(ds, max_iter=200)
        zc = np.load(os.path.join(cache, "Match", "YOHO_C", "200iters", "0-1.npz"))

        np.savez(os.path.join(GOLD, "chain.npz"),
                 K=K, pair_seed=3, wseed1=seed, wseed2=seed + 1,
                 eqv0_head=eqv0[:8], eqv1_head=eqv1[:8],
                 eqv0_rowsum=eqv0.astype(np.float64).sum(axis=(1, 2)),
                 eqv1_rowsum=eqv1.astype(np.float64).sum(axis=(1, 2)),
                 inv0=np.mean(eqv0, axis=-1), inv1=np.mean(eqv1, axis=-1),
                 match=pps, dr_index=dr, cor=cor, quat16=q, trans_pre=trans_pre,
                 yohoo_trans=zo["trans"], yohoo_recall=int(zo["recalltime"]),
                 yohoo20_trans=zo20["trans"], yohoo20_recall=int(zo20["recalltime"]),
                 yohoc_trans=zc["trans"], yohoc_recall=int(zc["recalltime"]), yohoc_center=zc["center"],
                 prelog_o=np.array(prelog_o))
        print(f"chain: K={K} matches={pps.shape[0]} dr_true={pair['gi']} dr_hist_top={np.bincount(dr).argmax()} "
              f"yohoo_recall={int(zo['recalltime'])} yohoc_recall={int(zc['recalltime'])}")

        # ---------------- 10. r_eval.matrix_from_quaternion on raw fp32 quats --------------
        qs = (W.hash_uniform(5, "quat", 64 * 4).reshape(64, 4) - 0.5).astype(np.float32)
        qs /= np.linalg.norm(qs, axis=1, keepdims=True).astype(np.float32)
        mats = np.stack([r_eval.matrix_from_quaternion(qq) for qq in qs])
        np.savez(os.path.join(GOLD, "quat.npz"), q=qs, mats=mats)

        # ---------------- 11. Threepps2Tran incl. LAPACK reflection sign -------------------
        rs = np.random.RandomState(8)
        k0 = rs.rand(50, 3, 3) * 3
        k1 = rs.rand(50, 3, 3) * 3
        Ts = np.stack([est_c.Threepps2Tran(k0[i], k1[i]) for i in range(50)])
        np.savez(os.path.join(GOLD, "kabsch.npz"), k0=k0, k1=k1, T=Ts)
        # ---------------- 12. evaluator: FMR + Registration Recall on a 4-fragment scene -----------
        import tests.evaluator as evaluator      # imports utils.RR_cal (nibabel stub) and utils.dataset (open3d stub)
        import utils.dataset as rdataset
        import utils.RR_cal as rr
        sc = make_scene(4, 64, seed=21, tables=tb)
        sroot = os.path.join(work, "origin", "synth4", "room")
        sname = "synth4/room"
        scache = os.path.join(work, "cache", "Testset", sname)
        write_scene_files(sc, sroot, scache)
        ds4 = rdataset.ThrDMatchPartDataset(sroot, 4)
        ds4.name = sname
        # the reference's get_kps needs the fragment .ply (open3d) or trips over an undefined attribute
        # (utils/dataset.py:107); the synthetic scene only has Keypoints_PC/*.npy - the file the hot path reads.
        ds4.get_kps = lambda cid: np.load(ds4.kps_pc_fn[int(cid)])
        datasets = {"wholesetname": "synth4", "room": ds4}
        outs = {}
        for part, Ev, it, sign, seedv in (("PartI", evaluator.Evaluator_PartI, 100, "YOHO_C", 5), ("PartII", evaluator.Evaluator_PartII, 1000, "YOHO_O", 6)):
            c = cfg(part)
            c.extractor, c.matcher, c.estimator, c.descriptor = part, "Match", ("yohoc" if part == "PartI" else "yohoo"), "YOHO"
            c.fmr_ratio, c.ok_match_dist_threshold, c.RR_dist_threshold, c.testset_name = 0.05, 0.1, 0.2, "synth4"
            ev = Ev(c, it)
            np.random.seed(seedv)
            ev.run_onescene(ds4)
            FMR, pair_fmrs = ev.Feature_match_Recall(ds4, ratio=c.fmr_ratio)
            RR, c_flags, c_errors = rr.benchmark(c, datasets, it, yoho_sign=sign)
            pre = open(os.path.join(scache, "Match", sign, f"{it}iters", "pre.log")).read()
            res_txt = open(os.path.join(work, "cache", "Testset", "synth4", "Eval_results", f"{sign}_RR", f"{it}iters", "result.txt")).read()
            outs[part] = dict(FMR=FMR, pair_fmrs=pair_fmrs, RR=RR, flags=np.array(c_flags[sname]), errors=np.array(c_errors[sname]),
                              prelog=np.array(pre), result_txt=np.array(res_txt))
            print(f"evaluator {part}: FMR={FMR:.3f} RR={RR:.3f} flags={c_flags[sname]}")
        matches = {f"match_{a}_{b}": np.load(os.path.join(scache, "Match", f"{a}-{b}.npy")) for (a, b) in sc["pairs"]}
        gtp = rdataset.ThrDMatchPartDataset.parse_gt_fn(os.path.join(sroot, "PointCloud", "gt.log"))
        np.savez(os.path.join(GOLD, "scene4.npz"), seed=21, K=64, nfrag=4,
                 gt_log=np.array(open(os.path.join(sroot, "PointCloud", "gt.log")).read()),
                 gt_info=np.array(open(os.path.join(sroot, "PointCloud", "gt.info")).read()),
                 parsed_keys=np.array(sorted(gtp.keys())), parsed_T=np.stack([gtp[k] for k in sorted(gtp.keys())]),
                 **{f"{part}_{k}": v for part, o in outs.items() for k, v in o.items()}, **matches)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print("golden fixtures written to", GOLD, {f: os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD)})


if __name__ == "__main__":
    main()
