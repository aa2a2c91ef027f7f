"""CPU tests for the SURVEY section 8(f) "next" rows #1/#2: the Registration-Recall benchmark (utils/RR_cal.py) and
the evaluation-dataset description (utils/dataset.py), against outputs of the reference itself
(tests/golden/scene4.npz, produced by oracle/gen_golden.py)."""
import os
import types
import numpy as np
import pytest

from yoho_amd import RR_cal, synth
from yoho_amd.dataset import ThrDMatchPartDataset, get_dataset_name, read_ply_xyz


def _lay_out(tmp_path, g):
    root = tmp_path / "origin" / "synth4" / "room"
    os.makedirs(root / "PointCloud")
    (root / "PointCloud" / "gt.log").write_text(str(g["gt_log"]))
    (root / "PointCloud" / "gt.info").write_text(str(g["gt_info"]))
    return root


def test_scene_generator_reproduces_golden_inputs(gold, tmp_path):
    g = gold("scene4.npz")
    sc = synth.make_scene(int(g["nfrag"]), int(g["K"]), seed=int(g["seed"]))
    synth.write_scene_files(sc, str(tmp_path / "s"))
    assert (tmp_path / "s" / "PointCloud" / "gt.log").read_text() == str(g["gt_log"])
    assert (tmp_path / "s" / "PointCloud" / "gt.info").read_text() == str(g["gt_info"])


def test_parse_gt_and_dataset(gold, tmp_path):
    g = gold("scene4.npz")
    root = _lay_out(tmp_path, g)
    parsed = ThrDMatchPartDataset.parse_gt_fn(str(root / "PointCloud" / "gt.log"))
    assert sorted(parsed.keys()) == list(g["parsed_keys"])
    for k, T in zip(g["parsed_keys"], g["parsed_T"]):
        assert parsed[k].dtype == np.float32 and np.array_equal(parsed[k], T)
    ds = ThrDMatchPartDataset(str(root), 4)
    assert ds.pc_ids == ["0", "1", "2", "3"] and ds.pair_ids[0] == ("0", "1") and ds.get_pair_nums() == 6
    assert np.array_equal(ds.get_transform("1", "3"), parsed["1-3"])
    os.makedirs(root / "Keypoints_PC")
    k = np.random.RandomState(0).rand(7, 3)
    np.save(root / "Keypoints_PC" / "cloud_bin_2Keypoints.npy", k)
    assert np.array_equal(ds.get_kps("2"), k)
    with pytest.raises(NotImplementedError):
        get_dataset_name("nope", "/x")


def test_3dLomatch_dataset_shares_3dmatch_files(gold, tmp_path):
    """utils/dataset.py:163-182: '3dLomatch' is 3dmatch's scene directories with gtLo.log as the ground truth and its own name
    (which the stage classes map back to 3dmatch's caches, tests/extractor.py:84-87)."""
    from yoho_amd.dataset import _SCENES
    from yoho_amd.utils import dataset_feature_name
    g = gold("sceneLo.npz")
    lo_pairs = [tuple(int(v) for v in p) for p in g["lo_pairs"]]
    sc = synth.make_scene(int(g["nfrag"]), int(g["K"]), seed=int(g["seed"]), res_deg=[float(v) for v in g["res_deg"]])
    scenes, counts = _SCENES["3dLomatch"]
    assert (scenes, counts) == _SCENES["3dmatch"] and counts == [60, 60, 60, 55, 57, 37, 66, 38]
    for s in scenes:
        synth.write_scene_files(sc, str(tmp_path / "3dmatch" / s), lo_pairs=lo_pairs)
    d3, dlo = get_dataset_name("3dmatch", str(tmp_path)), get_dataset_name("3dLomatch", str(tmp_path))
    assert dlo["wholesetname"] == "3dLomatch" and list(dlo)[1:] == scenes
    for s, n in zip(scenes, counts):
        a, b = d3[s], dlo[s]
        assert b.name == f"3dLomatch/{s}" and a.name == f"3dmatch/{s}" and dataset_feature_name(b.name) == a.name
        assert b.root == a.root == f"{tmp_path}/3dmatch/{s}" and b.kps_pc_fn == a.kps_pc_fn and len(b.pc_ids) == n
        assert b.gt_dir.endswith("PointCloud/gtLo.log") and a.gt_dir.endswith("PointCloud/gt.log")
        assert [tuple(int(v) for v in p) for p in b.pair_ids] == lo_pairs and len(a.pair_ids) == 15
        assert np.array_equal(b.get_transform("1", "4"), a.get_transform("1", "4"))
    assert dataset_feature_name("3dmatch/kitchen") == "3dmatch/kitchen" and dataset_feature_name("ETH/wood_summer") == "ETH/wood_summer"


@pytest.mark.parametrize("part,sign,it", [("PartI", "YOHO_C", 100), ("PartII", "YOHO_O", 1000)])
def test_benchmark_on_3dLomatch_matches_reference(gold, tmp_path, part, sign, it):
    """RR_cal.benchmark for a '3dLomatch/..' scene reads gtLo.log / gtLo.info (utils/RR_cal.py:340-353: the dataset's gt_dir with the
    extension swapped) and the pre.log under Testset/3dLomatch/..; flags, RR, errors and result.txt of the reference's run."""
    g = gold("sceneLo.npz")
    lo_pairs = [tuple(int(v) for v in p) for p in g["lo_pairs"]]
    sc = synth.make_scene(int(g["nfrag"]), int(g["K"]), seed=int(g["seed"]), res_deg=[float(v) for v in g["res_deg"]])
    root = tmp_path / "origin" / "3dmatch" / "room"
    synth.write_scene_files(sc, str(root), lo_pairs=lo_pairs)
    cache = tmp_path / "cache"
    pre_dir = cache / "Testset" / "3dLomatch/room" / "Match" / sign / f"{it}iters"
    os.makedirs(pre_dir)
    (pre_dir / "pre.log").write_text(str(g[f"{part}_prelog"]))
    ds = ThrDMatchPartDataset(str(root), int(g["nfrag"]), f"{root}/PointCloud/gtLo.log")
    ds.name = "3dLomatch/room"
    cfg = types.SimpleNamespace(output_cache_fn=str(cache), RR_dist_threshold=0.2)
    RR, flags, errors = RR_cal.benchmark(cfg, {"wholesetname": "3dLomatch", "room": ds}, it, yoho_sign=sign)
    assert RR == float(g[f"{part}_RR"]) and flags["3dLomatch/room"] == list(g[f"{part}_flags"])
    assert np.allclose(errors["3dLomatch/room"], g[f"{part}_errors"], rtol=1e-9, atol=1e-12)
    txt = (cache / "Testset" / "3dLomatch" / "Eval_results" / f"{sign}_RR" / f"{it}iters" / "result.txt").read_text()
    assert txt == str(g[f"{part}_result_txt"])


@pytest.mark.parametrize("fixture", ["sceneWHU.npz", "sceneETH.npz"])
@pytest.mark.parametrize("part,sign", [("PartI", "YOHO_C"), ("PartII", "YOHO_O")])
def test_benchmark_on_whu_tls_and_eth_matches_reference(gold, tmp_path, fixture, part, sign):
    """BASELINE config 5's other two test sets.  'WHU-TLS' switches utils/RR_cal.benchmark to the CONSECUTIVE protocol
    (utils/RR_cal.py:329-331 -> :262-285: every gt row counts, the first estimate is scored against gt row 0 directly,
    n_gt = table entries + 1); 'ETH' keeps the 3DMatch protocol (the pair that is gt row 0 can never be looked up: flag 2).
    Golden: the reference's benchmark on the pre.log its own estimators wrote (oracle/gen_golden_r5.py) - flags, RR, errors, result.txt."""
    g = gold(fixture)
    whole, scene, it = str(g["whole"]), str(g["scene"]), int(g["iters"])
    pairs = [tuple(int(v) for v in p) for p in g["pairs"]]
    sc = synth.make_scene(int(g["nfrag"]), int(g["K"]), seed=int(g["seed"]), res_deg=[float(v) for v in g["res_deg"]])
    sc["pairs"] = pairs
    scenes, counts = {"WHU-TLS": (['Park', 'Mountain', 'Campus', 'RiverBank', 'UndergroundExcavation', 'Tunnel'], [32, 6, 10, 7, 12, 7]),
                      "ETH": (['gazebo_summer', 'gazebo_winter', 'wood_autumn', 'wood_summer'], [32, 31, 32, 37])}[whole]   # utils/dataset.py:195-208
    for s in scenes:
        synth.write_scene_files(sc, str(tmp_path / "origin" / whole / s))
    dss = get_dataset_name(whole, str(tmp_path / "origin"))
    assert dss["wholesetname"] == whole and list(dss)[1:] == scenes and [len(dss[s].pc_ids) for s in scenes] == counts
    ds = dss[scene]
    assert ds.name == f"{whole}/{scene}" and [tuple(int(v) for v in p) for p in ds.pair_ids] == pairs
    cache = tmp_path / "cache"
    pre_dir = cache / "Testset" / ds.name / "Match" / sign / f"{it}iters"
    os.makedirs(pre_dir)
    (pre_dir / "pre.log").write_text(str(g[f"{part}_prelog"]))
    cfg = types.SimpleNamespace(output_cache_fn=str(cache), RR_dist_threshold=0.2)
    RR, flags, errors = RR_cal.benchmark(cfg, {"wholesetname": whole, scene: ds}, it, yoho_sign=sign)
    assert RR == float(g[f"{part}_RR"]) and flags[ds.name] == list(g[f"{part}_flags"])
    assert np.allclose(errors[ds.name], g[f"{part}_errors"], rtol=1e-9, atol=1e-12)
    txt = (cache / "Testset" / whole / "Eval_results" / f"{sign}_RR" / f"{it}iters" / "result.txt").read_text()
    assert txt == str(g[f"{part}_result_txt"])
    if whole == "WHU-TLS":
        assert 2 not in flags[ds.name] and len(flags[ds.name]) == len(pairs)       # consecutive pairs are all scored, row 0 included
        # the same pre.log under the 3DMatch protocol would score nothing: every pair is consecutive
        ds3 = types.SimpleNamespace(name=ds.name, gt_dir=ds.gt_dir)
        with np.errstate(invalid="ignore"):
            rr3, f3, _ = RR_cal.benchmark(cfg, {"wholesetname": "other", scene: ds3}, it, yoho_sign=sign)
        assert f3[ds.name] == [2] * len(pairs) and np.isnan(rr3)                   # 0 / 0 as in the reference (numpy integer n_gt)
    else:
        assert flags[ds.name][0] == 2                                              # gt row 0: table entry 0 = "empty" (:250-256)


def test_ply_reader(tmp_path):
    pts = np.random.RandomState(1).rand(11, 3).astype(np.float32)
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex 11\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nend_header\n"
    rec = np.zeros(11, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1")])
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    (tmp_path / "a.ply").write_bytes(hdr.encode() + rec.tobytes())
    assert np.array_equal(read_ply_xyz(str(tmp_path / "a.ply")), pts.astype(np.float64))
    asc = "ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n1 2 3\n4 5 6\n"
    (tmp_path / "b.ply").write_text(asc)
    assert np.array_equal(read_ply_xyz(str(tmp_path / "b.ply")), [[1, 2, 3], [4, 5, 6]])


@pytest.mark.parametrize("part,sign,it", [("PartI", "YOHO_C", 100), ("PartII", "YOHO_O", 1000)])
def test_benchmark_matches_reference(gold, tmp_path, part, sign, it):
    g = gold("scene4.npz")
    root = _lay_out(tmp_path, g)
    cache = tmp_path / "cache"
    pre_dir = cache / "Testset" / "synth4/room" / "Match" / sign / f"{it}iters"
    os.makedirs(pre_dir)
    (pre_dir / "pre.log").write_text(str(g[f"{part}_prelog"]))
    ds = types.SimpleNamespace(name="synth4/room", gt_dir=str(root / "PointCloud" / "gt.log"))
    cfg = types.SimpleNamespace(output_cache_fn=str(cache), RR_dist_threshold=0.2)
    RR, flags, errors = RR_cal.benchmark(cfg, {"wholesetname": "synth4", "room": ds}, it, yoho_sign=sign)
    assert RR == float(g[f"{part}_RR"])
    assert flags["synth4/room"] == list(g[f"{part}_flags"])
    assert np.allclose(errors["synth4/room"], g[f"{part}_errors"], rtol=1e-9, atol=1e-12)
    txt = (cache / "Testset" / "synth4" / "Eval_results" / f"{sign}_RR" / f"{it}iters" / "result.txt").read_text()
    assert txt == str(g[f"{part}_result_txt"])


def test_rr_helpers():
    assert np.allclose(RR_cal.mat2quat(np.eye(3)), [1, 0, 0, 0])
    R = np.array([[0., -1, 0], [1, 0, 0], [0, 0, 1]])
    assert np.allclose(RR_cal.mat2quat(R), [np.sqrt(.5), 0, 0, np.sqrt(.5)])
    assert np.allclose(RR_cal.rotation_error(np.eye(3)[None], R[None]), 90.0)
    assert np.allclose(RR_cal.translation_error(np.zeros((1, 3, 1)), np.ones((1, 3, 1))), np.sqrt(3))
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [1, 2, 3]
    assert np.isclose(RR_cal.computeTransformationErr(T, np.eye(6)), 14 + 0.5)
