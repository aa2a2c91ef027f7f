"""GPU tests (-m gpu): the drop-in classes (same names / cfg / .npy cache layout as the reference's
tests/extractor.py, tests/matcher.py, tests/estimator.py, simple_yoho/yoho_extract.py) against the
golden vectors produced by the reference's own classes on the same synthetic pair."""
import os
import sys
import types
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import yoho_oracle as orc  # noqa: E402
from yoho_amd import synth, weights as W  # noqa: E402

pytestmark = pytest.mark.gpu


class FakeDataset:
    """duck type of utils/dataset.py:ThrDMatchPartDataset"""
    def __init__(self, name, pc_ids, pair_ids, kps, gt):
        self.name, self.pc_ids, self.pair_ids, self._kps, self._gt = name, pc_ids, pair_ids, kps, gt

    def get_kps(self, pc_id):
        return self._kps[pc_id]

    def get_transform(self, id0, id1):
        return self._gt


def _make_workdir(tmp_path, gold, sd1, sd2, golden):
    from yoho_amd import store
    store.clear()
    g = gold(golden)
    pr = synth.make_pair(int(g["K"]), seed=int(g["pair_seed"]))
    model_fn = tmp_path / "model"
    for sub, sd in (("PartI_train", sd1), ("PartII_train", sd2)):
        os.makedirs(model_fn / sub)
        W.save_checkpoint(str(model_fn / sub / "model_best.pth"), sd, 0.5)
    name = "synth/room"
    cache = tmp_path / "cache" / "Testset" / name
    os.makedirs(cache / "FCGF_Input_Group_feature")
    np.save(cache / "FCGF_Input_Group_feature" / "0.npy", pr["feat0"])
    np.save(cache / "FCGF_Input_Group_feature" / "1.npy", pr["feat1"])
    kdir = tmp_path / "origin" / name / "Keypoints_PC"
    os.makedirs(kdir)
    np.save(kdir / "cloud_bin_0Keypoints.npy", pr["keys0"])
    np.save(kdir / "cloud_bin_1Keypoints.npy", pr["keys1"])
    ds = FakeDataset(name, ["0", "1"], [("0", "1")], {"0": pr["keys0"], "1": pr["keys1"]}, pr["gt"])

    def cfg(part):
        return types.SimpleNamespace(
            SO3_related_files=None, model_fn=str(model_fn), output_cache_fn=str(tmp_path / "cache"),
            origin_data_dir=str(tmp_path / "origin"), test_network_type=f"{part}_test", train_network_type=f"{part}_train",
            test_batch_size=40 if part == "PartI" else 50, ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09)
    return types.SimpleNamespace(cfg=cfg, ds=ds, cache=str(cache), gold=g, pair=pr)


@pytest.fixture()
def workdir(tmp_path, gold, sd1, sd2):
    return _make_workdir(tmp_path, gold, sd1, sd2, "chain.npz")


def prelog_numbers(text):
    """every number of a pre.log (headers and matrix rows), in file order"""
    return np.array([float(v) for v in str(text).split()])


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(np.max(np.abs(b)), 1e-30))


def test_full_chain_matches_reference_outputs(workdir):
    from yoho_amd import extractor, matcher, estimator
    w, g = workdir, workdir.gold
    assert set(extractor.name2extractor) == {"PartI", "PartII"} and set(matcher.name2matcher) == {"Match"}
    assert set(estimator.name2estimator) == {"yohoc", "yohoc_mul", "yohoo"}

    extractor.name2extractor["PartI"](w.cfg("PartI")).Extract(w.ds)
    eqv0 = np.load(f"{w.cache}/YOHO_Output_Group_feature/0.npy")
    eqv1 = np.load(f"{w.cache}/YOHO_Output_Group_feature/1.npy")
    assert eqv0.dtype == np.float32 and eqv0.shape == (int(g["K"]), 32, 60)
    assert rel(eqv0[:8], g["eqv0_head"]) < 1e-4 and rel(eqv1[:8], g["eqv1_head"]) < 1e-4
    assert np.allclose(eqv0.astype(np.float64).sum(axis=(1, 2)), g["eqv0_rowsum"], atol=1e-3)

    matcher.name2matcher["Match"](w.cfg("PartI")).match(w.ds)
    pps = np.load(f"{w.cache}/Match/0-1.npy")
    assert pps.dtype == np.int64 and np.array_equal(pps, g["match"])            # bit-exact indices

    extractor.extractor_dr_index(w.cfg("PartI")).PartI_Rindex(w.ds)
    dr = np.load(f"{w.cache}/Match/DR_index/0-1.npy")
    assert dr.dtype == np.int64 and np.array_equal(dr, g["dr_index"])

    extractor.name2extractor["PartII"](w.cfg("PartII")).PartII_R_pre(w.ds)
    T = np.load(f"{w.cache}/Match/Trans_pre/0-1.npy")
    assert T.dtype == np.float64 and T.shape == g["trans_pre"].shape
    assert rel(T, g["trans_pre"]) < 1e-4                                         # R,t within 1e-4 relative

    np.random.seed(1234)
    estimator.name2estimator["yohoo"](w.cfg("PartII")).ransac(w.ds, max_iter=1000)
    z = np.load(f"{w.cache}/Match/YOHO_O/1000iters/0-1.npz")
    assert int(z["recalltime"]) == int(g["yohoo_recall"]) and rel(z["trans"], g["yohoo_trans"]) < 1e-4
    log = open(f"{w.cache}/Match/YOHO_O/1000iters/pre.log").read()
    ref = str(g["prelog_o"])
    assert log.split("\n")[0] == ref.split("\n")[0] and len(log.split("\n")) == len(ref.split("\n"))
    assert rel(prelog_numbers(log), prelog_numbers(ref)) < 1e-4                  # every numeric row of the trajectory file
    np.random.seed(4321)
    estimator.yohoo(w.cfg("PartII")).ransac(w.ds, max_iter=20)
    z = np.load(f"{w.cache}/Match/YOHO_O/20iters/0-1.npz")
    assert int(z["recalltime"]) == int(g["yohoo20_recall"]) and rel(z["trans"], g["yohoo20_trans"]) < 1e-4

    for cls in ("yohoc", "yohoc_mul"):
        np.random.seed(99)
        estimator.name2estimator[cls](w.cfg("PartI")).ransac(w.ds, max_iter=200)
        z = np.load(f"{w.cache}/Match/YOHO_C/200iters/0-1.npz")
        assert int(z["recalltime"]) == int(g["yohoc_recall"])
        assert np.allclose(z["trans"], g["yohoc_trans"], rtol=0, atol=1e-9)
        assert np.allclose(z["center"], g["yohoc_center"], rtol=0, atol=0)


def test_config1_256_keypoint_chain(tmp_path, gold, sd1, sd2):
    """BASELINE.json config 1 at its stated size (a 256-keypoint pair): the drop-in stage classes against the outputs of
    the reference's own classes on the same inputs (oracle/gen_golden_r2.py), pre.log compared number by number"""
    from yoho_amd import extractor, matcher, estimator
    w = _make_workdir(tmp_path, gold, sd1, sd2, "chain256.npz")
    g = w.gold
    extractor.name2extractor["PartI"](w.cfg("PartI")).Extract(w.ds)
    eqv0 = np.load(f"{w.cache}/YOHO_Output_Group_feature/0.npy")
    eqv1 = np.load(f"{w.cache}/YOHO_Output_Group_feature/1.npy")
    assert eqv0.shape == (256, 32, 60) and eqv0.dtype == np.float32
    rows = g["rows"]
    assert rel(eqv0[rows], g["eqv0_rows"]) < 1e-4 and rel(eqv1[rows], g["eqv1_rows"]) < 1e-4
    assert np.allclose(eqv0.astype(np.float64).sum(axis=(1, 2)), g["eqv0_rowsum"], atol=1e-3)
    assert np.allclose(eqv1.astype(np.float64).sum(axis=(1, 2)), g["eqv1_rowsum"], atol=1e-3)
    matcher.name2matcher["Match"](w.cfg("PartI")).match(w.ds)
    assert np.array_equal(np.load(f"{w.cache}/Match/0-1.npy"), g["match"])
    extractor.extractor_dr_index(w.cfg("PartI")).PartI_Rindex(w.ds)
    dr = np.load(f"{w.cache}/Match/DR_index/0-1.npy")
    gap = g["cor_top2"][:, 1] - g["cor_top2"][:, 0]
    differ = dr != g["dr_index"]
    print("config 1: %d matches, %d Des2R near-ties (gap < 1e-4), %d disagreements" % (len(dr), int((gap < 1e-4).sum()), int(differ.sum())))
    assert (gap[differ] < 1e-4).all() and differ.sum() <= 1
    if differ.any():                                   # continue on the reference's indices so that later stages stay comparable
        np.save(f"{w.cache}/Match/DR_index/0-1.npy", g["dr_index"])
    extractor.name2extractor["PartII"](w.cfg("PartII")).PartII_R_pre(w.ds)
    assert rel(np.load(f"{w.cache}/Match/Trans_pre/0-1.npy"), g["trans_pre"]) < 1e-4
    np.random.seed(1234)
    estimator.name2estimator["yohoo"](w.cfg("PartII")).ransac(w.ds, max_iter=1000)
    z = np.load(f"{w.cache}/Match/YOHO_O/1000iters/0-1.npz")
    assert int(z["recalltime"]) == int(g["yohoo_recall"]) and rel(z["trans"], g["yohoo_trans"]) < 1e-4
    assert rel(prelog_numbers(open(f"{w.cache}/Match/YOHO_O/1000iters/pre.log").read()), prelog_numbers(g["prelog_o"])) < 1e-4
    np.random.seed(99)
    estimator.name2estimator["yohoc"](w.cfg("PartI")).ransac(w.ds, max_iter=300)
    z = np.load(f"{w.cache}/Match/YOHO_C/300iters/0-1.npz")
    assert int(z["recalltime"]) == int(g["yohoc_recall"])
    assert np.allclose(z["trans"], g["yohoc_trans"], rtol=0, atol=1e-9) and np.array_equal(z["center"], g["yohoc_center"])
    assert np.allclose(prelog_numbers(open(f"{w.cache}/Match/YOHO_C/300iters/pre.log").read()), prelog_numbers(g["prelog_c"]), rtol=0, atol=1e-9)


def test_yohoc_device_sampling_mode(workdir):
    """cfg.yohoc_device_sampling: statistic, sampling, Kabsch and vote on the device.  The triples are those of the oracle's
    restatement of the device sampler (bit-exact), the winner and transform those of the oracle's loop over them."""
    from yoho_amd import estimator, extractor, matcher
    w = workdir
    extractor.name2extractor["PartI"](w.cfg("PartI")).Extract(w.ds)
    matcher.name2matcher["Match"](w.cfg("PartI")).match(w.ds)
    extractor.extractor_dr_index(w.cfg("PartI")).PartI_Rindex(w.ds)
    c = w.cfg("PartI")
    c.yohoc_device_sampling = True
    est = estimator.yohoc(c)
    np.random.seed(77)
    seed = estimator.draw_seed()
    np.random.seed(77)
    est.ransac(w.ds, max_iter=500)
    z = np.load(f"{w.cache}/Match/YOHO_C/500iters/0-1.npz")
    pps, dr = np.load(f"{w.cache}/Match/0-1.npy"), np.load(f"{w.cache}/Match/DR_index/0-1.npy")
    k0, k1 = w.pair["keys0"][pps[:, 0]], w.pair["keys1"][pps[:, 1]]
    tri = orc.yohoc_device_triples(dr, 500, seed)
    it, cnt, T, _ = orc.yohoc_select(k0, k1, tri, 0.07, proper=True)
    assert int(z["recalltime"]) == it and np.allclose(z["trans"], T, rtol=0, atol=1e-9)
    assert np.array_equal(z["center"], np.concatenate([k0[tri[it - 1]], k1[tri[it - 1]]], 0))
    # the library's sampled triples, bit-exact against the oracle's restatement, and the per-iteration votes
    ctx = est.ctx
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _, res, tri_d = ctx.c_ransac_device(cu(w.pair["keys0"]), cu(w.pair["keys1"]), cu(dr), 500, seed, 0.07, match=cu(pps), want_triples=True)
    assert np.array_equal(tri_d.cpu().numpy(), tri) and tuple(res.cpu().numpy()) == (it, cnt)
    # no bucket with two matches -> the reference's "no estimate" code
    uniq = np.arange(min(len(dr), 60), dtype=np.int64)           # every match its own coarse rotation
    _, res, _ = ctx.c_ransac_device(cu(k0[:len(uniq)]), cu(k1[:len(uniq)]), cu(uniq), 50, 1, 0.07)
    assert tuple(res.cpu().numpy()) == (50001, 0) and orc.yohoc_device_triples(uniq, 50, 1) is None


def test_stage_skip_if_cached_and_missing_model(workdir, tmp_path):
    from yoho_amd import extractor
    w = workdir
    ex = extractor.extractor_PartI(w.cfg("PartI"))
    ex.Extract(w.ds)
    fn = f"{w.cache}/YOHO_Output_Group_feature/0.npy"
    t0 = os.path.getmtime(fn)
    ex.Extract(w.ds)                               # existing outputs are skipped, not overwritten
    assert os.path.getmtime(fn) == t0
    bad = w.cfg("PartI"); bad.model_fn = str(tmp_path / "nope")
    with pytest.raises(ValueError, match="No model exists"):
        extractor.extractor_PartI(bad).Extract(w.ds)


def test_network_and_knn_mirrors(workdir, sd1, tables):
    from yoho_amd.network import name2network
    from yoho_amd.knn_search import knn_module
    net = name2network["PartI_test"](workdir.cfg("PartI")).cuda()
    net.load_state_dict(sd1, strict=True)
    net.eval()
    x = synth.unit_features(1, seed=4)                       # B == 1 crashes the reference; must work here
    o = net(torch.from_numpy(x).cuda())
    e, i = orc.partI_forward(x, sd1, tables.N)
    assert rel(o["eqv"].cpu().numpy(), e) < 1e-4 and rel(o["inv"].cpu().numpy(), i) < 1e-4
    with pytest.raises(Exception):
        net.load_state_dict({"bogus": np.zeros(3, np.float32)}, strict=True)
    knn = knn_module.KNN(1)
    a = np.random.RandomState(0).randn(50, 32).astype(np.float32)
    b = np.random.RandomState(1).randn(70, 32).astype(np.float32)
    d, idx = knn(torch.from_numpy(b.T[None]).cuda(), torch.from_numpy(a.T[None]).cuda())   # (target, source)
    assert d.shape == (1, 1, 50) and idx.shape == (1, 1, 50)
    ref = orc.pdist_l2(a, b)
    assert np.array_equal(idx[0, 0].numpy(), ref.argmin(1)) and np.array_equal(d[0, 0].numpy(), ref.min(1))


def test_yoho_extractor_with_stub_backbone(sd1, tables):
    from yoho_amd.yoho_extract import yoho_extractor

    class StubFCGF:               # any object with run(pc, voxel_size) can stand in for the backbone
        def run(self, pc, voxel_size):
            ds = pc[::3].astype(np.float32)
            rs = np.random.RandomState(len(ds))
            f = rs.randn(len(ds), 32).astype(np.float32)
            return ds, f / np.linalg.norm(f, axis=1, keepdims=True)

    rs = np.random.RandomState(0)
    pc = rs.rand(3000, 3)
    ex = yoho_extractor(yoho_ckpt=sd1, fcgf=StubFCGF())
    np.random.seed(5)
    kpts, inv, eqv = ex.run(pc, voxel_size=0.025, nkpts=200)
    assert kpts.shape == (200, 3) and tuple(inv.shape) == (200, 32) and tuple(eqv.shape) == (200, 32, 60)
    assert not inv.is_cuda and not eqv.is_cuda
    # oracle: same sampling, same stub, brute-force transfer, PartI
    np.random.seed(5)
    kidx = np.random.permutation(len(pc))[0:200]
    feats = np.empty((200, 32, 60), np.float32)
    for g in range(60):
        kr = (pc[kidx] @ tables.R64[g].T).astype(np.float32)
        ds, f = StubFCGF().run(pc @ tables.R64[g].T, 0.025)
        j = np.argmin(orc.pdist_l2(kr, ds, squared=True), 1)
        feats[:, :, g] = f[j]
    e, i = orc.partI_forward(feats, sd1, tables.N)
    assert np.array_equal(kpts, pc[kidx]) and rel(eqv.numpy(), e) < 1e-4 and rel(inv.numpy(), i) < 1e-4
    with pytest.raises(NotImplementedError):
        yoho_extractor(fcgf_ckpt=None, yoho_ckpt=sd1).run(pc)


def test_extractor_outputs_stop_being_pinned_beyond_the_budget(sd1, monkeypatch):
    """yoho_extractor.run returns page-locked CPU tensors only while the pinned bytes still alive in the caller's hands stay under
    yoho_extract.PIN_OUTPUT_BYTES (a caller that keeps a scene's descriptors must not pin GBs of host RAM); values are the
    same either way, dropping results gives the budget back."""
    import gc
    from yoho_amd import yoho_extract as ye

    class Stub:
        feats = {}                                                     # the 60 rotated copies of a call share one feature table (360 calls here)

        def run(self, pc, voxel_size):
            if len(pc) not in self.feats:
                f = torch.from_numpy(synth.unit_features(len(pc), seed=1)[:, :, 0].copy())
                self.feats[len(pc)] = f / f.norm(dim=1, keepdim=True)
            return pc, self.feats[len(pc)]
    pc = synth.surface_cloud(600, seed=2)
    ex = ye.yoho_extractor(fcgf_ckpt=None, yoho_ckpt=sd1, fcgf=Stub())
    gc.collect()
    base = ye._pinned_alive[0]
    per_call = 200 * 32 * 61 * 4                                   # inv (200,32) + eqv (200,32,60)
    monkeypatch.setattr(ye, "PIN_OUTPUT_BYTES", base + 2 * per_call + 64)
    kept = []
    for i in range(4):
        np.random.seed(3)
        kept.append(ex.run(pc, voxel_size=0.025, nkpts=200))
    assert [k[2].is_pinned() for k in kept] == [True, True, False, False] and [k[1].is_pinned() for k in kept] == [True, True, False, False]
    for k in kept[1:]:
        assert torch.equal(k[1], kept[0][1]) and torch.equal(k[2], kept[0][2])
    assert ye._pinned_alive[0] == base + 2 * per_call
    del kept, k
    gc.collect()
    assert ye._pinned_alive[0] == base
    np.random.seed(3)
    assert ex.run(pc, voxel_size=0.025, nkpts=200)[2].is_pinned()
    monkeypatch.setattr(ye, "PIN_OUTPUT_BYTES", 0)
    np.random.seed(3)
    assert not ex.run(pc, voxel_size=0.025, nkpts=200)[2].is_pinned()


def test_fcgf_extractor_dropin_and_full_yoho_extractor(sd1, tables):
    """simple_yoho/fcgf_feat.py + yoho_extract.py with the HIP backbone: checkpoint dict in the FCGF format"""
    import fcgf_oracle as fo
    from yoho_amd.fcgf_feat import fcgf_extractor
    from yoho_amd.yoho_extract import yoho_extractor
    fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
    ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7},
          "state_dict": {k: torch.from_numpy(np.array(v)) for k, v in fsd.items()}}
    pc = synth.surface_cloud(2500, seed=3)
    fx = fcgf_extractor(ck)
    ds, feat = fx.run(pc, voxel_size=0.025)
    sel, F0 = fo.extract_features(pc, 0.025, fsd)
    assert np.array_equal(ds, pc[sel]) and not feat.is_cuda and rel(feat.numpy(), F0) < 1e-4
    # full extractor = the same parts chained over the 60 rotations (reference simple_yoho/yoho_extract.py:41-60)
    ex = yoho_extractor(fcgf_ckpt=ck, yoho_ckpt=sd1)
    np.random.seed(7)
    kpts, inv, eqv = ex.run(pc, voxel_size=0.025, nkpts=64)
    np.random.seed(7)
    kidx = np.random.permutation(len(pc))[0:64]
    assert np.array_equal(kpts, pc[kidx]) and tuple(eqv.shape) == (64, 32, 60)
    feats = np.empty((64, 32, 60), np.float32)
    for g in (0, 17, 59):                                    # oracle on three of the sixty rotations
        pcg = pc @ tables.R64[g].T
        selg, Fg = fo.extract_features(pcg, 0.025, fsd)
        j = np.argmin(orc.pdist_l2((pc[kidx] @ tables.R64[g].T).astype(np.float32), pcg[selg].astype(np.float32), squared=True), 1)
        feats[:, :, g] = Fg[j]
    got = ex._last_group_feats.cpu().numpy()
    for g in (0, 17, 59):
        assert rel(got[:, :, g], feats[:, :, g]) < 1e-4, g
    e, i = orc.partI_forward(got, sd1, tables.N)
    assert rel(eqv.numpy(), e) < 1e-4 and rel(inv.numpy(), i) < 1e-4
    # the two-lane pipeline (backbone passes alternating over two streams / library contexts, the default) changes no bit: the same
    # call with every pass on the caller's stream (another split of the sixty rotations into passes is NOT bit-identical: the kernel
    # variants of a pass follow its row count, NOTEBOOK 9.6 - 8e-7 here)
    assert ex.lanes == 2 and ex._side_stream is not None
    for lanes, rb in ((1, 15),):
        ex.lanes, ex.rot_batch = lanes, rb
        np.random.seed(7)
        k1, i1, e1 = ex.run(pc, voxel_size=0.025, nkpts=64)
        assert torch.equal(ex._last_group_feats.cpu(), torch.from_numpy(got)), (lanes, rb, float((ex._last_group_feats.cpu() - torch.from_numpy(got)).abs().max()))
        assert np.array_equal(k1, kpts) and torch.equal(i1, inv) and torch.equal(e1, eqv), (lanes, rb)
    ex.lanes, ex.rot_batch = 2, 15
    # streamed over several clouds (descriptor pass and result copy of a fragment on a tail lane while the backbone lanes work on the
    # next one): the values run() returns, fragment by fragment, with the generator consumed in the same order
    pcs = [pc, synth.surface_cloud(1900, seed=4), pc[:2000]]
    np.random.seed(11)
    one = [ex.run(p, voxel_size=0.025, nkpts=48) for p in pcs]
    np.random.seed(11)
    many = list(ex.run_many(pcs, voxel_size=0.025, nkpts=48))
    assert len(many) == 3
    for a, b in zip(one, many):
        assert np.array_equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert list(ex.run_many([])) == []
    # leaving the stream early (queued fragments still running on the lanes) leaves nothing behind that the next call could trip over
    np.random.seed(11)
    g = ex.run_many(pcs, voxel_size=0.025, nkpts=48)
    first = next(g)
    g.close()
    assert np.array_equal(first[0], one[0][0]) and torch.equal(first[2], one[0][2])
    np.random.seed(11)
    again = ex.run(pcs[0], voxel_size=0.025, nkpts=48)
    assert np.array_equal(again[0], one[0][0]) and torch.equal(again[1], one[0][1]) and torch.equal(again[2], one[0][2])


@pytest.mark.parametrize("golden", ["scene4.npz", "scene6.npz"])
@pytest.mark.parametrize("part,it,seedv", [("PartI", 100, 5), ("PartII", 1000, 6)])
def test_evaluator_fmr_and_registration_recall(gold, tmp_path, sd1, sd2, part, it, seedv, golden):
    """Evaluator_PartI/II.eval on a synthetic scene reproduces the reference's FMR, per-pair flags and
    Registration Recall (golden: tests/evaluator.py + utils/RR_cal.py run on the reference).  scene4: 4 fragments, random
    PartII head (YOHO-O fails everywhere: RR 0); scene6: 6 fragments with planted small / large residual rotations and the
    near-identity quaternion head, so that YOHO-O registers some pairs and not others (RR 0.6)."""
    from yoho_amd import evaluator, store
    from yoho_amd.dataset import ThrDMatchPartDataset
    store.clear()
    g = gold(golden)
    if golden == "scene6.npz":
        sd2 = W.identity_head(sd2)
        sc = synth.make_scene(int(g["nfrag"]), int(g["K"]), seed=int(g["seed"]), res_deg=[float(v) for v in g["res_deg"]])
        assert 0.0 < float(g["PartII_RR"]) < 1.0
    else:
        sc = synth.make_scene(int(g["nfrag"]), int(g["K"]), seed=int(g["seed"]))
    nfrag = int(g["nfrag"])
    sroot = tmp_path / "origin" / "synth4" / "room"
    cache = tmp_path / "cache"
    synth.write_scene_files(sc, str(sroot), str(cache / "Testset" / "synth4/room"))
    model_fn = tmp_path / "model"
    for sub, sd in (("PartI_train", sd1), ("PartII_train", sd2)):
        os.makedirs(model_fn / sub)
        W.save_checkpoint(str(model_fn / sub / "model_best.pth"), sd, 0.5)
    ds = ThrDMatchPartDataset(str(sroot), nfrag)
    ds.name = "synth4/room"
    datasets = {"wholesetname": "synth4", "room": ds}

    def cfg(p):
        return types.SimpleNamespace(
            SO3_related_files=None, model_fn=str(model_fn), output_cache_fn=str(cache), origin_data_dir=str(tmp_path / "origin"),
            test_network_type=f"{p}_test", train_network_type=f"{p}_train", test_batch_size=40 if p == "PartI" else 50,
            ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09, extractor=p, matcher="Match",
            estimator="yohoc" if p == "PartI" else "yohoo", descriptor="YOHO", fmr_ratio=0.05,
            ok_match_dist_threshold=0.1, RR_dist_threshold=0.2, testset_name="synth4")
    if part == "PartII":      # the PartII evaluator assumes the PartI descriptors are already cached (tests/evaluator.py:112-117)
        from yoho_amd import extractor
        extractor.extractor_PartI(cfg("PartI")).Extract(ds)
    ev = evaluator.name2evaluator[part](cfg(part), it)
    np.random.seed(seedv)
    RR, FMRS, pair_fmrs = ev.eval(datasets, results_log=str(tmp_path / "results.log"))
    for (a, b) in sc["pairs"]:
        assert np.array_equal(np.load(cache / "Testset" / "synth4/room" / "Match" / f"{a}-{b}.npy"), g[f"match_{a}_{b}"])
    assert np.array_equal(pair_fmrs, g[f"{part}_pair_fmrs"]) and FMRS[0] == float(g[f"{part}_FMR"])
    assert RR == float(g[f"{part}_RR"])
    print("%s %s: RR %.3f (reference %.3f), FMR %.3f" % (golden, part, RR, float(g[f"{part}_RR"]), FMRS[0]))
    sign = "YOHO_C" if part == "PartI" else "YOHO_O"
    from yoho_amd import RR_cal
    _, mine = RR_cal.read_pre_trajectory(str(cache / "Testset" / "synth4/room" / "Match" / sign / f"{it}iters" / "pre.log"))
    ref_log = tmp_path / "ref_pre.log"
    ref_log.write_text(str(g[f"{part}_prelog"]))
    _, ref = RR_cal.read_pre_trajectory(str(ref_log))
    assert rel(mine, ref) < 1e-4
    assert "Mean_Registration_Recall" in (tmp_path / "results.log").read_text()


def _lay_out_lo(tmp_path, g, sd1, sd2):
    """sceneLo.npz's scene as the reference saw it: origin/3dmatch/room (keypoints, gt.log, gtLo.log / gtLo.info), the FCGF group
    features under Testset/3dmatch/room, PartII weights with the near-identity quaternion head."""
    from yoho_amd import store
    from yoho_amd.dataset import ThrDMatchPartDataset
    store.clear()
    nfrag = int(g["nfrag"])
    lo_pairs = [tuple(int(v) for v in p) for p in g["lo_pairs"]]
    sc = synth.make_scene(nfrag, int(g["K"]), seed=int(g["seed"]), res_deg=[float(v) for v in g["res_deg"]])
    sroot = tmp_path / "origin" / "3dmatch" / "room"
    cache = tmp_path / "cache"
    synth.write_scene_files(sc, str(sroot), str(cache / "Testset" / "3dmatch/room"), lo_pairs=lo_pairs)
    model_fn = tmp_path / "model"
    for sub, sd in (("PartI_train", sd1), ("PartII_train", W.identity_head(sd2))):
        os.makedirs(model_fn / sub)
        W.save_checkpoint(str(model_fn / sub / "model_best.pth"), sd, 0.5)
    ds3 = ThrDMatchPartDataset(str(sroot), nfrag)
    ds3.name = "3dmatch/room"
    dsLo = ThrDMatchPartDataset(str(sroot), nfrag, f"{sroot}/PointCloud/gtLo.log")        # utils/dataset.py:176-182
    dsLo.name = "3dLomatch/room"
    assert [tuple(int(v) for v in p) for p in dsLo.pair_ids] == lo_pairs and len(ds3.pair_ids) == nfrag * (nfrag - 1) // 2

    def cfg(p):
        return types.SimpleNamespace(
            SO3_related_files=None, model_fn=str(model_fn), output_cache_fn=str(cache), origin_data_dir=str(tmp_path / "origin"),
            test_network_type=f"{p}_test", train_network_type=f"{p}_train", test_batch_size=40 if p == "PartI" else 50,
            ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09, extractor=p, matcher="Match",
            estimator="yohoc" if p == "PartI" else "yohoo", descriptor="YOHO", fmr_ratio=0.05,
            ok_match_dist_threshold=0.1, RR_dist_threshold=0.2, testset_name="3dLomatch")
    return types.SimpleNamespace(sc=sc, ds3=ds3, dsLo=dsLo, cfg=cfg, cache=cache, lo_pairs=lo_pairs,
                                 datasets={"wholesetname": "3dLomatch", "room": dsLo})


@pytest.mark.parametrize("part,it,seedv,sign", [("PartI", 100, 15, "YOHO_C"), ("PartII", 1000, 16, "YOHO_O")])
def test_3dLomatch_evaluators_match_reference(gold, tmp_path, sd1, sd2, part, it, seedv, sign):
    """BASELINE config 4's 3DLoMatch half - the name-mapping path (tests/extractor.py:84-87,:152-158, tests/matcher.py:24-27,
    tests/evaluator.py:41-47,:50-54, tests/estimator.py:84-87,:310-313, utils/dataset.py:176-182): a dataset named '3dLomatch/room'
    reads features, descriptors and keypoints of '3dmatch/room', writes its own Match / DR_index / Trans_pre / YOHO_* / pre.log under
    Testset/3dLomatch/room, and Evaluator_PartI skips Extract for it.  Golden: the reference's own classes on the same files
    (oracle/gen_golden_r4.py): match lists, coarse rotations, FMR, per-pair flags, RR, pre.log, result.txt."""
    from yoho_amd import evaluator, extractor, RR_cal
    g = gold("sceneLo.npz")
    w = _lay_out_lo(tmp_path, g, sd1, sd2)
    cache3, cacheLo = w.cache / "Testset" / "3dmatch/room", w.cache / "Testset" / "3dLomatch/room"
    ev = evaluator.name2evaluator[part](w.cfg(part), it)
    if part == "PartI":
        # without 3dmatch's descriptors the 3dLomatch evaluation has nothing to read: Extract is NOT run for '3dLo' names
        with pytest.raises(FileNotFoundError):
            ev.run_onescene(w.dsLo)
        import shutil
        shutil.rmtree(cacheLo)
    extractor.extractor_PartI(w.cfg("PartI")).Extract(w.ds3)          # what a 3dmatch evaluation leaves behind
    np.random.seed(seedv)
    RR, FMRS, pair_fmrs = ev.eval(w.datasets, results_log=str(tmp_path / "results.log"))
    assert sorted(os.listdir(cache3)) == ["FCGF_Input_Group_feature", "YOHO_Output_Group_feature"]
    assert sorted(os.listdir(cacheLo)) == ["Match"]
    for (a, b) in w.lo_pairs:
        assert np.array_equal(np.load(cacheLo / "Match" / f"{a}-{b}.npy"), g[f"match_{a}_{b}"])
        assert np.array_equal(np.load(cacheLo / "Match" / "DR_index" / f"{a}-{b}.npy"), g[f"dr_{a}_{b}"])
        if part == "PartII":
            z = np.load(cacheLo / "Match" / "YOHO_O" / "1000iters" / f"{a}-{b}.npz")
            assert int(z["recalltime"]) == int(g[f"yohoo_recall_{a}_{b}"]) and rel(z["trans"], g[f"yohoo_trans_{a}_{b}"]) < 1e-4
    assert np.array_equal(pair_fmrs, g[f"{part}_pair_fmrs"]) and FMRS[0] == float(g[f"{part}_FMR"])
    assert RR == float(g[f"{part}_RR"])
    _, flags, errors = RR_cal.benchmark(w.cfg(part), w.datasets, it, yoho_sign=sign)
    assert flags["3dLomatch/room"] == list(g[f"{part}_flags"])
    txt = (w.cache / "Testset" / "3dLomatch" / "Eval_results" / f"{sign}_RR" / f"{it}iters" / "result.txt").read_text()
    assert txt.splitlines()[0] == str(g[f"{part}_result_txt"]).splitlines()[0] and "3dLomatch/room" in txt
    _, mine = RR_cal.read_pre_trajectory(str(cacheLo / "Match" / sign / f"{it}iters" / "pre.log"))
    ref_log = tmp_path / "ref_pre.log"
    ref_log.write_text(str(g[f"{part}_prelog"]))
    ref_pairs, ref = RR_cal.read_pre_trajectory(str(ref_log))
    assert rel(mine, ref) < 1e-4
    print("sceneLo %s: RR %.3f (reference %.3f), FMR %.3f" % (part, RR, float(g[f"{part}_RR"]), FMRS[0]))


def test_3dLomatch_through_the_dataset_driver(gold, tmp_path, sd1, sd2):
    """The sharded driver on the same 3DLoMatch files: run_dataset.eval_sharded reads 3dmatch's FCGF features / keypoints for a
    '3dLomatch/..' scene, writes npz + pre.log under Testset/3dLomatch/.., and gets the reference's YOHO-O flags and RR."""
    from yoho_amd import run_dataset, RR_cal
    g = gold("sceneLo.npz")
    w = _lay_out_lo(tmp_path, g, sd1, sd2)
    stats = {}
    rr = run_dataset.eval_sharded(w.cfg("PartII"), max_iter=1000, estimator="yohoo", datasets=w.datasets, base_seed=3,
                                  results_log=str(tmp_path / "results.log"), stats_out=stats)
    _, flags, _ = RR_cal.benchmark(w.cfg("PartII"), w.datasets, 1000, yoho_sign="YOHO_O")
    assert flags["3dLomatch/room"] == list(g["PartII_flags"]) and rr == float(g["PartII_RR"])
    assert stats["pairs"] == len(w.lo_pairs) and stats["fragments"] == int(g["nfrag"])
    sdir = w.cache / "Testset" / "3dLomatch/room" / "Match" / "YOHO_O" / "1000iters"
    est_pairs, _ = RR_cal.read_pre_trajectory(str(sdir / "pre.log"))
    assert [tuple(int(v) for v in p[:2]) for p in est_pairs] == w.lo_pairs
    assert not (w.cache / "Testset" / "3dmatch/room" / "Match").exists()


def _lay_out_r5(tmp_path, g, sd1, sd2):
    """sceneWHU.npz / sceneETH.npz as the reference saw them (oracle/gen_golden_r5.py): origin/{whole}/{scene} with the fixture's pair
    list as gt.log, FCGF group features under Testset/{whole}/{scene}, PartII weights with the near-identity quaternion head."""
    from yoho_amd import store
    from yoho_amd.dataset import ThrDMatchPartDataset
    store.clear()
    whole, scene = str(g["whole"]), str(g["scene"])
    pairs = [tuple(int(v) for v in p) for p in g["pairs"]]
    nfrag = int(g["nfrag"])
    sc = synth.make_scene(nfrag, int(g["K"]), seed=int(g["seed"]), res_deg=[float(v) for v in g["res_deg"]])
    sc["pairs"] = pairs
    sroot = tmp_path / "origin" / whole / scene
    cache = tmp_path / "cache"
    synth.write_scene_files(sc, str(sroot), str(cache / "Testset" / f"{whole}/{scene}"))
    model_fn = tmp_path / "model"
    for sub, sd in (("PartI_train", sd1), ("PartII_train", W.identity_head(sd2))):
        os.makedirs(model_fn / sub)
        W.save_checkpoint(str(model_fn / sub / "model_best.pth"), sd, 0.5)
    ds = ThrDMatchPartDataset(str(sroot), nfrag)       # utils/dataset.py:195-215 builds the set's scenes like this (the table itself: tests/test_rr_cpu.py)
    ds.name = f"{whole}/{scene}"
    assert ds.name == f"{whole}/{scene}" and [tuple(int(v) for v in p) for p in ds.pair_ids] == pairs

    def cfg(p):
        return types.SimpleNamespace(
            SO3_related_files=None, model_fn=str(model_fn), output_cache_fn=str(cache), origin_data_dir=str(tmp_path / "origin"),
            test_network_type=f"{p}_test", train_network_type=f"{p}_train", test_batch_size=40 if p == "PartI" else 50,
            ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09, extractor=p, matcher="Match",
            estimator="yohoc" if p == "PartI" else "yohoo", descriptor="YOHO", fmr_ratio=0.05,
            ok_match_dist_threshold=0.1, RR_dist_threshold=0.2, testset_name=whole)
    return types.SimpleNamespace(sc=sc, ds=ds, cfg=cfg, cache=cache, pairs=pairs, whole=whole, scene=scene, name=ds.name,
                                 datasets={"wholesetname": whole, scene: ds})


@pytest.mark.parametrize("fixture", ["sceneWHU.npz", "sceneETH.npz"])
@pytest.mark.parametrize("part,sign", [("PartI", "YOHO_C"), ("PartII", "YOHO_O")])
def test_whu_tls_and_eth_evaluators_match_reference(gold, tmp_path, sd1, sd2, fixture, part, sign):
    """BASELINE config 5's ETH / WHU-TLS halves at the reference's own settings (1000 iterations).  Evaluator_PartI swaps in
    `yohoc_mul` above 500 iterations (tests/evaluator.py:37-38): the reference forks one process per pair, so EVERY pair draws from
    the np.random state the parent had (tests/estimator.py:255-275) - several pairs here, where yohoc and yohoc_mul differ; golden =
    the files the reference's Pool wrote.  'WHU-TLS' is scored by the consecutive protocol (utils/RR_cal.py:329-331, :262-285), 'ETH'
    by 3DMatch's.  Compared: match lists, coarse rotations, per-pair transforms / winning iteration / winning triple, FMR, flags, RR,
    pre.log, result.txt."""
    from yoho_amd import evaluator, extractor, estimator, RR_cal
    g = gold(fixture)
    w = _lay_out_r5(tmp_path, g, sd1, sd2)
    it = int(g["iters"])
    cdir = w.cache / "Testset" / w.name
    ev = evaluator.name2evaluator[part](w.cfg(part), it)
    if part == "PartI":
        assert type(ev.estimator) is estimator.yohoc_mul and type(evaluator.Evaluator_PartI(w.cfg(part), 500).estimator) is estimator.yohoc
    else:
        extractor.extractor_PartI(w.cfg("PartI")).Extract(w.ds)       # the PartII evaluator assumes cached PartI descriptors
    np.random.seed(int(g["seed_c" if part == "PartI" else "seed_o"]))
    state = np.random.get_state()
    RR, FMRS, pair_fmrs = ev.eval(w.datasets, results_log=str(tmp_path / "results.log"))
    if part == "PartI":                                               # the forks leave the parent's stream where it was
        assert all(np.array_equal(a, b) for a, b in zip(np.random.get_state()[1:], state[1:]))
    for (a, b) in w.pairs:
        assert np.array_equal(np.load(cdir / "Match" / f"{a}-{b}.npy"), g[f"match_{a}_{b}"])
        assert np.array_equal(np.load(cdir / "Match" / "DR_index" / f"{a}-{b}.npy"), g[f"dr_{a}_{b}"])
        key = "yohoc" if part == "PartI" else "yohoo"
        z = np.load(cdir / "Match" / sign / f"{it}iters" / f"{a}-{b}.npz")
        assert int(z["recalltime"]) == int(g[f"{key}_recall_{a}_{b}"]), (a, b)
        assert rel(z["trans"], g[f"{key}_trans_{a}_{b}"]) < 1e-4
        if part == "PartI":
            assert np.array_equal(z["center"], g[f"yohoc_center_{a}_{b}"])       # the winning triple's six keypoints
    assert np.array_equal(pair_fmrs, g[f"{part}_pair_fmrs"]) and FMRS[0] == float(g[f"{part}_FMR"])
    assert RR == float(g[f"{part}_RR"])
    _, flags, errors = RR_cal.benchmark(w.cfg(part), w.datasets, it, yoho_sign=sign)
    assert flags[w.name] == list(g[f"{part}_flags"])
    assert np.allclose(errors[w.name], g[f"{part}_errors"], rtol=1e-3, atol=1e-6)
    txt = (w.cache / "Testset" / w.whole / "Eval_results" / f"{sign}_RR" / f"{it}iters" / "result.txt").read_text()
    ref_txt = str(g[f"{part}_result_txt"])
    assert txt.splitlines()[0] == ref_txt.splitlines()[0] and w.name in txt
    assert [ln for ln in txt.splitlines() if "Registration Recall" in ln] == [ln for ln in ref_txt.splitlines() if "Registration Recall" in ln]
    _, mine = RR_cal.read_pre_trajectory(str(cdir / "Match" / sign / f"{it}iters" / "pre.log"))
    ref_log = tmp_path / "ref_pre.log"
    ref_log.write_text(str(g[f"{part}_prelog"]))
    _, ref = RR_cal.read_pre_trajectory(str(ref_log))
    assert rel(mine, ref) < 1e-4
    print("%s %s: RR %.3f (reference %.3f), FMR %.3f" % (fixture, part, RR, float(g[f"{part}_RR"]), FMRS[0]))


@pytest.mark.parametrize("fixture", ["sceneWHU.npz", "sceneETH.npz"])
def test_whu_tls_and_eth_through_the_dataset_driver(gold, tmp_path, sd1, sd2, fixture):
    """The sharded driver on the same files: run_dataset.eval_sharded (YOHO-O, and YOHO-C with the sampling on the device) on a
    'WHU-TLS/..' / 'ETH/..' scene gets the reference's flags and RR under that set's protocol, pre.log in gt.log's pair order."""
    from yoho_amd import run_dataset, RR_cal
    g = gold(fixture)
    it = int(g["iters"])
    for estimator, part, sign in (("yohoo", "PartII", "YOHO_O"), ("yohoc", "PartI", "YOHO_C")):
        w = _lay_out_r5(tmp_path / estimator, g, sd1, sd2)
        stats = {}
        rr = run_dataset.eval_sharded(w.cfg(part), max_iter=it, estimator=estimator, datasets=w.datasets, base_seed=3,
                                      results_log=str(tmp_path / "results.log"), stats_out=stats)
        _, flags, _ = RR_cal.benchmark(w.cfg(part), w.datasets, it, yoho_sign=sign)
        assert flags[w.name] == list(g[f"{part}_flags"]) and rr == float(g[f"{part}_RR"]), (estimator, flags[w.name])
        assert stats["pairs"] == len(w.pairs) and stats["fragments"] == int(g["nfrag"])
        sdir = w.cache / "Testset" / w.name / "Match" / sign / f"{it}iters"
        est_pairs, _ = RR_cal.read_pre_trajectory(str(sdir / "pre.log"))
        assert [tuple(int(v) for v in p[:2]) for p in est_pairs] == w.pairs
        txt = (w.cache / "Testset" / w.whole / "Eval_results" / f"{sign}_RR" / f"{it}iters" / "result.txt").read_text()
        assert txt.splitlines()[0] == str(g[f"{part}_result_txt"]).splitlines()[0]


def test_testset_create_from_point_clouds(tmp_path, tables):
    """YOHO_testset.py drop-in: fragment point clouds + keypoints -> FCGF_Input_Group_feature/{id}.npy; three of the
    sixty group elements are checked against the oracle chain (voxelise, backbone, f64 NN gather)."""
    import fcgf_oracle as fo
    from yoho_amd.YOHO_testset import testset_create
    fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
    ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7}, "state_dict": fsd}
    clouds = {"0": synth.surface_cloud(2200, seed=11), "1": synth.surface_cloud(1800, seed=12)}
    rs = np.random.RandomState(0)
    kps = {k: v[rs.permutation(len(v))[:40]] for k, v in clouds.items()}

    class DS:
        pc_ids = ["0", "1"]
        get_pc = staticmethod(lambda i: clouds[i])
        get_kps = staticmethod(lambda i: kps[i])

    cfg = types.SimpleNamespace(model=ck, voxel_size=0.025, dataset="synth", output_dir=str(tmp_path), origin_dir=str(tmp_path),
                                datasets={"wholesetname": "synth", "room": DS()})
    tc = testset_create(cfg)
    assert tc.lanes == 2                                    # default: backbone passes on two lanes, results written by the writer thread
    tc.batch_feature_extraction()
    assert tc.stats["fragments"] == 2
    # one lane (every pass on the caller's stream, rounds 1-5) writes the same bytes; the single-fragment API joins its lanes
    cfg1 = types.SimpleNamespace(**{**vars(cfg), "output_dir": str(tmp_path / "one_lane")})
    tc1 = testset_create(cfg1)
    tc1.lanes = 1
    tc1.batch_feature_extraction()
    for pid in ("0", "1"):
        a = np.load(f"{tmp_path}/Testset/synth/room/FCGF_Input_Group_feature/{pid}.npy")
        b = np.load(f"{tmp_path}/one_lane/Testset/synth/room/FCGF_Input_Group_feature/{pid}.npy")
        assert np.array_equal(a, b), pid
        assert np.array_equal(tc.fragment_group_features(clouds[pid], kps[pid]).cpu().numpy(), a), pid
    for pid in ("0", "1"):
        got = np.load(f"{tmp_path}/Testset/synth/room/FCGF_Input_Group_feature/{pid}.npy")
        assert got.shape == (40, 32, 60) and got.dtype == np.float32
        for g in (0, 23, 59):
            pcg = clouds[pid] @ tables.R64[g].T
            sel, Fg = fo.extract_features(pcg, 0.025, fsd)
            ref = orc.group_gather_one(kps[pid], pcg[sel].astype(np.float32), Fg, tables.R64[g])[0]
            assert rel(got[:, :, g], ref) < 1e-4, (pid, g)


def test_dataset_driver_from_point_clouds_equals_the_cache_route(tmp_path, sd1, tables):
    """run_dataset.ScenePairRunner(backbone=testset_create): the FCGF group features computed from the fragments' point clouds on the
    device (rotate, voxelise, backbone, NN gather) instead of read from FCGF_Input_Group_feature/*.npy - YOHO_testset.py's stage folded
    into the evaluation, nothing between the raw cloud and the registration result on disk.  Same per-pair results as the two-stage
    route (testset_create writes the cache, the driver reads it), bit for bit."""
    from yoho_amd import run_dataset, hip
    from yoho_amd.YOHO_testset import testset_create
    fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
    ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7}, "state_dict": fsd}
    base = synth.surface_cloud(4000, seed=21)
    rs = np.random.RandomState(2)
    clouds, kps = {}, {}
    for i in range(3):                                        # three overlapping views of one surface, each in its own frame
        Rg = tables.R64[(7 * i) % 60]
        sub = base[rs.permutation(len(base))[:2600]]
        clouds[str(i)] = sub @ Rg.T + rs.uniform(-0.2, 0.2, 3)
        kps[str(i)] = clouds[str(i)][rs.permutation(2600)[:96]]

    class DS:
        name = "synth/room"
        pc_ids = ["0", "1", "2"]
        pair_ids = [("0", "1"), ("0", "2"), ("1", "2")]
        get_pc = staticmethod(lambda i: clouds[i])
        get_kps = staticmethod(lambda i: kps[i])

    ds = DS()
    cfg = types.SimpleNamespace(output_cache_fn=str(tmp_path), ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09)
    tcfg = types.SimpleNamespace(model=ck, voxel_size=0.025, dataset="synth", output_dir=str(tmp_path), origin_dir=str(tmp_path),
                                 datasets={"wholesetname": "synth", "room": ds})
    ctx = hip.get_context()
    ctx.load_partI(sd1)
    tc = testset_create(tcfg, ctx=ctx)
    tc.batch_feature_extraction()                            # route A, stage 1: the cache files
    out = {}
    for route, backbone in (("cache", None), ("clouds", tc)):
        runner = run_dataset.ScenePairRunner(cfg, ctx, estimator="yohoc", max_iter=60, base_seed=5, pair_workers=1, backbone=backbone)
        runner.setup_scene(ds, ds.pair_ids)
        out[route] = [runner.run_pair(ds, p) for p in ds.pair_ids]
        assert runner.stats["fragments"] == 3 and runner.stats["pairs"] == 3
        assert (runner.stats["bytes_read"] == 0) == (route == "clouds") and (runner.stats["backbone_s"] > 0) == (route == "clouds")
    for a, b in zip(out["cache"], out["clouds"]):
        assert np.array_equal(a["trans"], b["trans"]) and a["recalltime"] == b["recalltime"] and a["matches"] == b["matches"] and a["inliers"] == b["inliers"]
    assert all(r["matches"] >= 3 for r in out["clouds"])


@pytest.mark.parametrize("estimator,part,it", [("yohoo", "PartII", 1000), ("yohoc", "PartI", 100)])
def test_eval_sharded_world1_on_scene6(gold, tmp_path, sd1, sd2, tables, estimator, part, it):
    """The dataset driver's GPU worker (run_dataset.eval_sharded: load + describe every fragment once, HBM-resident pairs, pre.log,
    RR) at world = 1 on the scene6 files.  YOHO-O: the reference's per-pair success flags and Registration Recall (0.6) from
    tests/evaluator.py + utils/RR_cal.py; YOHO-C with the sampling on the device (its own Philox stream): the same success flags as
    the reference's np.random YOHO-C (statistical parity: every pair registers, RR 1.0).  Every pair's result equals a plain
    pipeline.run_pair on separately described fragments with the same pair seed, and fragments are released after their last pair."""
    from yoho_amd import run_dataset, store, RR_cal, pipeline, hip
    from yoho_amd.dataset import ThrDMatchPartDataset
    store.clear()
    g = gold("scene6.npz")
    sd2i = W.identity_head(sd2)
    nfrag = int(g["nfrag"])
    sc = synth.make_scene(nfrag, int(g["K"]), seed=int(g["seed"]), res_deg=[float(v) for v in g["res_deg"]])
    sroot = tmp_path / "origin" / "synth4" / "room"
    cache = tmp_path / "cache"
    synth.write_scene_files(sc, str(sroot), str(cache / "Testset" / "synth4/room"))
    model_fn = tmp_path / "model"
    for sub, sd in (("PartI_train", sd1), ("PartII_train", sd2i)):
        os.makedirs(model_fn / sub)
        W.save_checkpoint(str(model_fn / sub / "model_best.pth"), sd, 0.5)
    ds = ThrDMatchPartDataset(str(sroot), nfrag)
    ds.name = "synth4/room"
    datasets = {"wholesetname": "synth4", "room": ds}
    cfg = types.SimpleNamespace(SO3_related_files=None, model_fn=str(model_fn), output_cache_fn=str(cache), origin_data_dir=str(tmp_path / "origin"),
                                ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09, RR_dist_threshold=0.2, testset_name="synth4")
    stats = {}
    rr = run_dataset.eval_sharded(cfg, max_iter=it, estimator=estimator, datasets=datasets, base_seed=3, results_log=str(tmp_path / "results.log"),
                                  stats_out=stats)
    sign = "YOHO_O" if estimator == "yohoo" else "YOHO_C"
    _, flags, _ = RR_cal.benchmark(cfg, datasets, it, yoho_sign=sign)
    assert np.array_equal(np.asarray(flags[ds.name]), g[f"{part}_flags"]), (flags[ds.name], g[f"{part}_flags"])
    assert rr == float(g[f"{part}_RR"])
    assert stats["fragments"] == nfrag and stats["pairs"] == len(ds.pair_ids) and stats["peak_resident_fragments"] == nfrag
    assert stats["range_guard"]["repeats_this_run"] == 0 and not stats["range_guard"]["partI_stays_bf16x3"]
    # the files the reference's estimator leaves: one npz per pair + pre.log in pair order
    sdir = cache / "Testset" / "synth4/room" / "Match" / sign / f"{it}iters"
    est_pairs, est = RR_cal.read_pre_trajectory(str(sdir / "pre.log"))
    assert [tuple(int(v) for v in p[:2]) for p in est_pairs] == [(int(a), int(b)) for a, b in ds.pair_ids]
    # pair by pair against pipeline.run_pair on separately described fragments
    ctx = hip.get_context()
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    desc = {}
    for k in range(nfrag):
        o = ctx.partI_forward(cu(sc["feats"][k]), want_inv=False, want_inv_np=True)
        desc[str(k)] = (cu(sc["feats"][k]), cu(sc["keys"][k]), {"eqv": o["eqv"], "inv_np": o["inv_np"]})
    for (a, b), res in zip(ds.pair_ids, stats["results"]["room"]):
        seed = run_dataset.pair_seed(3, ds.name, a, b)
        fa, ka, oa = desc[str(a)]
        fb, kb, ob = desc[str(b)]
        r = pipeline.run_pair(ctx, fa, fb, ka, kb, inlier_dist=0.09 if estimator == "yohoo" else 0.07, max_iter=it,
                              order_rng=np.random.RandomState(seed & 0xFFFFFFFF), eqv=(oa, ob), estimator=estimator, seed=seed)
        assert np.array_equal(np.asarray(r.trans, np.float64), res["trans"]) and int(r.best_h) == res["recalltime"], (a, b)
        assert np.array_equal(r.match.cpu().numpy(), g[f"match_{a}_{b}"])
        z = np.load(sdir / f"{a}-{b}.npz")
        assert np.array_equal(z["trans"], res["trans"]) and int(z["recalltime"]) == res["recalltime"]
    # every fragment was released after its last pair
    runner = run_dataset.ScenePairRunner(cfg, ctx, estimator=estimator, max_iter=it, base_seed=3)
    pairs = [tuple(p) for p in ds.pair_ids]
    runner.setup_scene(ds, pairs[:5])                       # pairs (0,1) ... (0,5): fragment 0 five times, the others once
    assert set(runner.frag) == {"0", "1", "2", "3", "4", "5"}
    runner.run_pair(ds, pairs[0])
    assert set(runner.frag) == {"0", "2", "3", "4", "5"}
    for p in pairs[1:5]:
        runner.run_pair(ds, p)
    assert runner.frag == {} and runner.uses == {}


def test_eval_sharded_counts_range_repeats_and_workers_go_sticky(gold, tmp_path, sd1, sd2):
    """A PartII checkpoint whose activations leave the fp16 planes (BN gamma x 20000) through the dataset driver: the one-call pair
    reports the flag, the pair is composed once more in bf16x3 (not a second fp16x2 attempt), after three such pairs a worker's
    PartII stays in bf16x3, and the run's stats say so; every transform equals run_pair with PartII in bf16x3 from the start."""
    import warnings
    from yoho_amd import run_dataset, store, pipeline, hip
    from yoho_amd.dataset import ThrDMatchPartDataset
    store.clear()
    g = gold("scene6.npz")
    sdx = {k: v.copy() for k, v in W.identity_head(sd2).items()}
    key = "PartII_SO3_Conv_layers.0.comb_layer_in.0.weight"
    sdx[key] = (sdx[key] * np.float32(20000.0)).astype(np.float32)
    nfrag = int(g["nfrag"])
    sc = synth.make_scene(nfrag, int(g["K"]), seed=int(g["seed"]), res_deg=[float(v) for v in g["res_deg"]])
    sroot, cache = tmp_path / "origin" / "synth4" / "room", tmp_path / "cache"
    synth.write_scene_files(sc, str(sroot), str(cache / "Testset" / "synth4/room"))
    ds = ThrDMatchPartDataset(str(sroot), nfrag)
    ds.name = "synth4/room"
    cfg = types.SimpleNamespace(SO3_related_files=None, model_fn=None, output_cache_fn=str(cache), origin_data_dir=str(tmp_path / "origin"),
                                ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09, RR_dist_threshold=0.2, testset_name="synth4")
    ctx = hip.Context()
    stats = {}
    with warnings.catch_warnings(record=True) as wrn:
        warnings.simplefilter("always")
        run_dataset.eval_sharded(cfg, max_iter=1000, estimator="yohoo", datasets={"wholesetname": "synth4", "room": ds}, base_seed=3,
                                 ctx=ctx, state_dicts=(sd1, sdx), stats_out=stats, hypotheses="all")
    rg = stats["range_guard"]
    npairs = len(ds.pair_ids)
    print("range guard through the dataset driver:", rg)
    assert 3 <= rg["repeats_this_run"] <= npairs and rg["partII_repeats_since_checkpoint"] == rg["repeats_this_run"]
    assert rg["partII_workers_staying_bf16x3"] >= 1 and not rg["partI_stays_bf16x3"]
    assert any("STAYS in bf16x3" in str(w.message) for w in wrn)
    wide = hip.Context()
    wide.load_partI(sd1)
    wide.load_partII(sdx)
    wide.set_partII_mode("bf16x3")
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    desc = {}
    for k in range(nfrag):
        o = wide.partI_forward(cu(sc["feats"][k]), want_inv=False, want_inv_np=True)
        desc[str(k)] = (cu(sc["feats"][k]), cu(sc["keys"][k]), {"eqv": o["eqv"], "inv_np": o["inv_np"]})
    for (a, b), res in zip(ds.pair_ids, stats["results"]["room"]):
        seed = run_dataset.pair_seed(3, ds.name, a, b)
        fa, ka, oa = desc[str(a)]
        fb, kb, ob = desc[str(b)]
        r = pipeline.run_pair(wide, fa, fb, ka, kb, inlier_dist=0.09, max_iter=1000, order_rng=np.random.RandomState(seed & 0xFFFFFFFF),
                              eqv=(oa, ob), seed=seed)
        assert np.array_equal(np.asarray(r.trans, np.float64), res["trans"]) and int(r.best_h) == res["recalltime"], (a, b)


def _world2_dataset_worker(rank, port, workdir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                      YOHO_DIST_BACKEND="gloo", YOHO_FORCE_DEVICE="0")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_dataset
    out = bench_dataset.run(nfrag=10, kp=600, span=4, estimator="yohoo", workdir=workdir, runs=1, max_iter=100)
    q.put((rank, out.get("trans_sha256"), [r["registration_recall"] for r in out["runs"]], out["pairs"]))
    torch.distributed.destroy_process_group()


def test_dataset_driver_world2_on_one_gpu_equals_world1(tmp_path):
    """BASELINE config 5's code path with real kernels: the sharded dataset driver at world = 2 (two processes, gloo for the host-side
    collectives, both on cuda:0 - a single-GPU box has no second device for RCCL) against world = 1 on the same synthetic scene: every
    pair's transform and recalltime identical (digest over all pairs), the same Registration Recall, every pair processed once.
    Exercises the shard plan (the scene is cut over both ranks), per-rank fragment loading, the result gather, the per-rank npz
    writes and rank 0's pre.log / RR evaluation."""
    import socket
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_dataset
    one = bench_dataset.run(nfrag=10, kp=600, span=4, estimator="yohoo", workdir=str(tmp_path / "w1"), runs=1, max_iter=100)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world2_dataset_worker, args=(r, port, str(tmp_path / "w2"), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert one["pairs"] == got[0][3] == 30 and one["trans_sha256"] is not None
    assert got[0][1] == one["trans_sha256"], "world-2 transforms differ from world-1"
    assert got[0][2] == [r["registration_recall"] for r in one["runs"]]


def test_dataset_driver_missing_or_truncated_cache_file_fails_cleanly(tmp_path, sd1):
    """ScenePairRunner.setup_scene on a cache with a truncated and a missing fragment file: the loader's error reaches the caller
    (no silent zeros, no hang) and the loader thread ends - with more fragments queued than the hand-over queue holds, so that a
    loader blocked on a full queue would stay behind."""
    import threading
    import time
    from yoho_amd import run_dataset, hip, store
    from yoho_amd.dataset import ThrDMatchPartDataset
    store.clear()
    nfrag = 12
    sc = synth.make_scene(nfrag, 64, seed=5)
    sroot = tmp_path / "origin" / "synth4" / "room"
    cache = tmp_path / "cache"
    synth.write_scene_files(sc, str(sroot), str(cache / "Testset" / "synth4/room"))
    ds = ThrDMatchPartDataset(str(sroot), nfrag)
    ds.name = "synth4/room"
    cfg = types.SimpleNamespace(SO3_related_files=None, model_fn=str(tmp_path / "model"), output_cache_fn=str(cache), origin_data_dir=str(tmp_path / "origin"),
                                ransac_c_inlinerdist=0.07, ransac_o_inlinerdist=0.09, RR_dist_threshold=0.2, testset_name="synth4")
    ctx = hip.get_context()
    ctx.load_partI(sd1)
    pairs = [tuple(p) for p in ds.pair_ids]
    fdir = cache / "Testset" / "synth4/room" / "FCGF_Input_Group_feature"
    whole = (fdir / "1.npy").read_bytes()
    for damage, exc in (("truncate", IOError), ("remove", FileNotFoundError)):
        if damage == "truncate":
            (fdir / "1.npy").write_bytes(whole[:len(whole) // 2])
        else:
            (fdir / "1.npy").unlink()
        before = threading.active_count()
        runner = run_dataset.ScenePairRunner(cfg, ctx, estimator="yohoc", max_iter=50, base_seed=1)
        with pytest.raises(exc):
            runner.setup_scene(ds, pairs)
        t0 = time.time()
        while threading.active_count() > before and time.time() - t0 < 5.0:
            time.sleep(0.05)
        assert threading.active_count() <= before, "the loader thread is still alive"
    # intact again: the same runner class describes the scene
    (fdir / "1.npy").write_bytes(whole)
    runner = run_dataset.ScenePairRunner(cfg, ctx, estimator="yohoc", max_iter=50, base_seed=1)
    runner.setup_scene(ds, pairs)
    assert len(runner.frag) == nfrag


@pytest.mark.parametrize("estimator", ["yohoo", "yohoc"])
def test_dataset_driver_overlapped_one_call_pairs_equal_sequential_staged(tmp_path, monkeypatch, estimator):
    """The dataset driver's fast configuration - pairs as single library calls (yoho_register_pair) on three worker contexts,
    running WHILE further fragments and the next scene part are loaded and described - against its plainest one: part by part,
    one worker, every pair composed from the staged entries in Python.  Several scenes of different sizes (a three-fragment scene included): every pair's transform and recalltime identical (digest over all
    pairs of all scenes), the same Registration Recall."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_dataset
    monkeypatch.setattr(bench_dataset, "PRESET_3DMATCH", [("a", 9, 20), ("b", 3, 3), ("c", 12, 40), ("d", 5, 10), ("e", 7, 9)])
    outs = []
    for kw in (dict(pair_workers=3, fused=True, overlap=True), dict(pair_workers=1, fused=False, overlap=False),
               dict(pair_workers=2, fused=True, overlap=True)):
        o = bench_dataset.run(kp=700, estimator=estimator, workdir=str(tmp_path / "ds"), runs=1, max_iter=200, preset="3dmatch", **kw)
        outs.append((o["trans_sha256"], o["runs"][0]["registration_recall"], o["pairs"]))
    assert outs[0][2] == 82
    assert outs[0] == outs[1] == outs[2], outs
