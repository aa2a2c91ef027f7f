"""FCGF backbone oracle (oracle/fcgf_oracle.py): the sparse (transposed) convolution restated from MinkowskiEngine's sources
is pinned against torch's dense conv3d / conv_transpose3d on densified inputs; structural checks of the U-Net."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import fcgf_oracle as fo  # noqa: E402
from yoho_amd import weights as W  # noqa: E402


def _cloud(n, grid, seed):
    rs = np.random.RandomState(seed)
    c = np.unique(rs.randint(0, grid, size=(n, 3)), axis=0).astype(np.int32)
    rs.shuffle(c)
    return c


def _dense(coords, feat, grid):
    d = np.zeros((feat.shape[1], grid, grid, grid), dtype=np.float32)       # [c][x][y][z]
    d[:, coords[:, 0], coords[:, 1], coords[:, 2]] = feat.T
    return torch.from_numpy(d)[None]


def _wt(Wk, K):
    # ME kernel index k = a + K*b + K*K*c (x fastest)  ->  torch [co][ci][a][b][c] on a tensor indexed [x][y][z]
    return torch.from_numpy(np.ascontiguousarray(Wk.reshape(K, K, K, Wk.shape[1], Wk.shape[2]).transpose(4, 3, 2, 1, 0)))


def test_conv_matches_dense_conv3d():
    rs = np.random.RandomState(0)
    grid, K = 12, 3
    c = _cloud(500, grid, 1)
    x = rs.randn(len(c), 5).astype(np.float32)
    Wk = rs.randn(K ** 3, 5, 7).astype(np.float32)
    out = fo.conv(x, c, c, Wk, K, 1)
    ref = TF.conv3d(_dense(c, x, grid), _wt(Wk, K), padding=K // 2)[0].numpy()
    assert np.allclose(out, ref[:, c[:, 0], c[:, 1], c[:, 2]].T, atol=2e-5)
    # kernel size 5 (conv1 family)
    W5 = rs.randn(125, 5, 3).astype(np.float32)
    out = fo.conv(x, c, c, W5, 5, 1)
    ref = TF.conv3d(_dense(c, x, grid), _wt(W5, 5), padding=2)[0].numpy()
    assert np.allclose(out, ref[:, c[:, 0], c[:, 1], c[:, 2]].T, atol=5e-5)


def test_strided_and_transposed_conv_match_dense():
    rs = np.random.RandomState(2)
    grid = 12
    c1 = _cloud(400, grid, 3)
    c2 = fo.stride_coords(c1, 2)
    assert (c2 % 2 == 0).all() and len(np.unique(c2, axis=0)) == len(c2)
    assert set(map(tuple, c2.tolist())) == set(map(tuple, (c1 // 2 * 2).tolist()))
    x = rs.randn(len(c1), 4).astype(np.float32)
    Wk = rs.randn(27, 4, 6).astype(np.float32)
    out = fo.conv(x, c1, c2, Wk, 3, 1)                                   # stride-2 convolution, offsets on the input stride
    ref = TF.conv3d(_dense(c1, x, grid), _wt(Wk, 3), padding=1, stride=2)[0].numpy()
    j = c2 // 2
    assert np.allclose(out, ref[:, j[:, 0], j[:, 1], j[:, 2]].T, atol=2e-5)
    # transposed: coarse (stride 2) -> the existing fine map, offsets on the fine stride
    y = rs.randn(len(c2), 6).astype(np.float32)
    Wt = rs.randn(27, 6, 3).astype(np.float32)
    up = fo.conv(y, c2, c1, Wt, 3, 1, transpose=True)
    dy = _dense(c2 // 2, y, grid // 2)
    wt = torch.from_numpy(np.ascontiguousarray(Wt.reshape(3, 3, 3, 6, 3).transpose(3, 4, 2, 1, 0)))    # [ci][co][a][b][c]
    ref = TF.conv_transpose3d(dy, wt, stride=2, padding=1, output_padding=1)[0].numpy()
    assert np.allclose(up, ref[:, c1[:, 0], c1[:, 1], c1[:, 2]].T, atol=2e-5)


def test_quantize_first_occurrence_and_unet_shapes():
    rs = np.random.RandomState(4)
    pc = rs.rand(3000, 3) * 0.6
    sel, coords = fo.voxelize(pc, 0.025)
    assert (np.diff(sel) > 0).all() and len(np.unique(coords, axis=0)) == len(coords)
    q = np.floor(pc / 0.025).astype(np.int32)
    first = {}
    for i, t in enumerate(map(tuple, q.tolist())):
        first.setdefault(t, i)
    assert sorted(first.values()) == sel.tolist()
    sd = W.synth_state_dict(W.FCGF_SPEC, 3)
    sel2, F = fo.extract_features(pc[:1200], 0.025, sd)
    assert F.shape == (len(sel2), 32) and np.isfinite(F).all()
    assert np.allclose(np.linalg.norm(F, axis=1), 1, atol=1e-5)
    # translation by whole multiples of the coarsest stride leaves the features unchanged (all maps shift together)
    _, F2 = fo.extract_features(pc[:1200] + 8 * 0.025 * np.array([1.0, -2.0, 3.0]), 0.025, sd)
    assert F2.shape == F.shape and np.abs(F - F2).max() < 1e-5


def test_minkowski_engine_known_answers():
    """The coordinate-map / kernel-map semantics the oracle restates, on the known-answer vectors of MinkowskiEngine's own
    tests (2-D cases embedded at z = 0: the 3-D kernel index of a 2-D index k with dz = 0 is k + 9)."""
    # region order: x fastest, offsets -1..1 around (1, -1)   (tests/cpp/kernel_region_cpu_test.py:22-42)
    offs = fo.kernel_offsets(3, 1)
    around = (np.array([1, -1, 0]) + offs[9:18])[:, :2].tolist()
    assert around == [[0, -2], [1, -2], [2, -2], [0, -1], [1, -1], [2, -1], [0, 0], [1, 0], [2, 0]]
    # kernel map direction: in = out + offset   (tests/cpp/kernel_region_cpu_test.py:100-116; the third output row is
    # another batch item there, i.e. another cloud here)
    cin = np.array([[1, -1, 0], [2, 1, 0]], dtype=np.int32)
    cout = np.array([[1, 0, 0], [1, 2, 0]], dtype=np.int32)
    m = fo.kernel_map(cin, cout, 3, 1)
    assert m.shape == (27, 2)
    assert m[9 + 1].tolist() == [0, -1]                         # in_maps[1] = [0], out_maps[1] = [0]
    assert m[9 + 2].tolist() == [-1, 1]                         # in_maps[2] = [1], out_maps[2] = [1]
    assert (m[:9] < 0).all() and (m[18:] < 0).all()             # nothing off the z = 0 plane
    # kernel size 1: identity on the shared coordinates   (:88-98)
    assert fo.kernel_map(cin, cin, 1, 1).tolist() == [[0, 1]]
    # strided maps   (tests/cpp/coordinate_map_cpu_test.py:95-125): two batch items = two clouds
    b0 = np.array([[1, 1, 0], [2, 1, 0], [1, 0, 0]], dtype=np.int32)
    b1 = np.array([[0, 3, 0], [0, 2, 0]], dtype=np.int32)
    assert len(fo.stride_coords(b0, 4)) + len(fo.stride_coords(b1, 4)) == 2
    assert len(fo.stride_coords(b0, 1)) + len(fo.stride_coords(b1, 1)) == 5
    neg = np.array([[-1, 0, 0], [-2, 0, 0], [1, 0, 0], [0, 0, 0]], dtype=np.int32)
    assert sorted(fo.stride_coords(neg, 2)[:, 0].tolist()) == [-2, 0]
    # negative coordinates, stride 2: -3..3 -> {-4, -2, 0, 2}   (tests/python/coordinate_manager.py:183-200)
    line = np.array([[v, 0, 0] for v in range(-3, 4)], dtype=np.int32)
    assert sorted(fo.stride_coords(line, 2)[:, 0].tolist()) == [-4, -2, 0, 2]
    # quantisation with collisions keeps one row per voxel   (tests/python/quantization.py:104-113)
    q = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 1, 0]])
    assert q[fo.sparse_quantize(q)].tolist() == [[0, 0, 0], [0, 1, 0]]


def _ml(c):
    """1-D / 2-D MinkowskiEngine test coordinates (batch, x[, y]) -> per-batch (x, y, 0) int32 arrays (a batch item = a cloud here)"""
    c = np.asarray(c, dtype=np.int32)
    out = {}
    for row in c:
        xyz = list(row[1:]) + [0] * (4 - len(row))
        out.setdefault(int(row[0]), []).append(xyz)
    return {b: np.array(v, dtype=np.int32) for b, v in out.items()}


def test_minkowski_engine_known_answers_round3():
    """The remaining known-answer vectors of MinkowskiEngine's own tests that touch semantics the FCGF backbone uses (the table in
    DESIGN.md 3.5 lists, per resunet.py layer, which vector pins which semantic).  Batch items are separate clouds here."""
    # insertion of duplicates keeps one row per coordinate: 3 of 4   (tests/cpp/coordinate_map_cpu_test.py:12-15)
    ins = np.array([[0, 1], [1, 2], [2, 3], [2, 3]])
    assert sum(len(fo.sparse_quantize(v)) for v in _ml(ins).values()) == 3
    # ... and the map answers a duplicate's query with the row of its FIRST occurrence: queries 1, 2, 3 are found and map to rows
    # 1, 2, 2; (-1,1) and (0,0) are absent   (tests/cpp/coordinate_map_cpu_test.py:47-65)
    rows = {tuple(r): i for i, r in reversed(list(enumerate(ins.tolist())))}
    q = [[-1, 1], [1, 2], [2, 3], [2, 3], [0, 0]]
    found = [(i, rows[tuple(v)]) for i, v in enumerate(q) if tuple(v) in rows]
    assert [i for i, _ in found] == [1, 2, 3] and [r for _, r in found] == [1, 2, 2]
    keep = np.concatenate([np.flatnonzero(ins[:, 0] == b)[fo.sparse_quantize(v)] for b, v in _ml(ins).items()])
    assert sorted(keep.tolist()) == [0, 1, 2]                  # the oracle's quantisation keeps exactly those first rows
    # ME.utils.unique_coordinate_map: 3 unique of [[0,0],[0,0],[0,1],[0,2]]   (tests/python/coordinate_manager.py:255-258)
    assert len(fo.sparse_quantize(np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [2, 0, 0]]))) == 3
    # stride 2 of {1, 2, 3, 3} -> 2 coordinates, tensor stride 2   (tests/cpp/coordinate_map_cpu_test.py:80-86)
    s = fo.stride_coords(np.array([[1, 0, 0], [2, 0, 0], [3, 0, 0], [3, 0, 0]], dtype=np.int32), 2)
    assert len(s) == 2 and (s % 2 == 0).all()
    # stride 4 of batch 0 {1, 1, 2, 2} and batch 1 {0, 0, 1}: one coordinate per batch item = 2   (tests/python/coordinate_manager.py:33-58)
    cm = np.array([[0, 1], [0, 1], [0, 2], [0, 2], [1, 0], [1, 0], [1, 1]])
    assert sum(len(fo.stride_coords(v[fo.sparse_quantize(v)], 4)) for v in _ml(cm).values()) == 2
    # SparseTensor construction with duplicated rows: data_loader's figure (tests/python/common.py:55-76, 8 points per batch item,
    # 2 items), rows 0 = 1 and 2 = 3 duplicated -> 16 - 2 = 14 rows   (tests/python/sparse_tensor.py:91-98)
    fig = ["   X   ", "  X X  ", " XXXXX "]
    pts = np.array([[i, j, 0] for i, r in enumerate(fig) for j, ch in enumerate(r) if ch != " "], dtype=np.int32)
    assert len(pts) == 8
    b0 = pts.copy(); b0[0] = b0[1]; b0[2] = b0[3]
    assert len(fo.sparse_quantize(b0)) + len(fo.sparse_quantize(pts)) == 14
    # the same figure through a stride-2, kernel-3 convolution (tests/python/kernel_map.py:40-78): 5 output coordinates per item
    # = floor(c / 2) * 2; by the map definition of the sources (in = out + offset, offsets -1..1 on the INPUT stride:
    # src/kernel_region.hpp:196-216, src/coordinate_map_manager.cpp:728-752, and the C++ known answers of
    # tests/cpp/kernel_region_cpu_test.py:100-116) the map has 13 pairs per item.  That python test asserts 16 for two items;
    # no reading of the sources reproduces 8 per item (it equals the number of INPUT points, i.e. a non-overlapping stride map),
    # so the figure is recorded here and the contradiction stated rather than hidden (oracle/fcgf_oracle.py header).
    out = fo.stride_coords(pts, 2)
    assert sorted(map(tuple, out[:, :2].tolist())) == [(0, 2), (0, 4), (2, 0), (2, 2), (2, 4)]
    m = fo.kernel_map(pts, out, 3, 1)
    assert int((m >= 0).sum()) == 13 and (m[:9] < 0).all() and (m[18:] < 0).all()
    # every input is covered at least once and the centre tap of an output that is itself an input point maps to it
    assert set(m[m >= 0].tolist()) == set(range(8))
    for o, c in enumerate(out.tolist()):
        hit = [i for i, p in enumerate(pts.tolist()) if p == c]
        assert (m[13, o] == hit[0]) if hit else (m[13, o] == -1)


def _oracle_triples(rows, cin, cout):
    """kernel map (K, Nout) of input rows -> sorted int32 (P, 7) array of (kernel index, input xyz, output xyz)"""
    out = []
    for k in range(rows.shape[0]):
        o = np.flatnonzero(rows[k] >= 0)
        if len(o):
            out.append(np.concatenate([np.full((len(o), 1), k, np.int32), cin[rows[k, o]], cout[o]], 1))
    t = np.concatenate(out).astype(np.int32) if out else np.zeros((0, 7), np.int32)
    return t[np.lexsort(t.T[::-1])]


def _oracle_levels(q):
    sel = fo.sparse_quantize(q)
    c = [np.asarray(q, dtype=np.int32)[sel]]
    for ts in (2, 4, 8):
        c.append(fo.stride_coords(c[-1], ts))
    return sel, c


def _oracle_maps(c):
    maps = {"conv1": (fo.kernel_map(c[0], c[0], 7, 1), c[0], c[0])}
    for l, ts in enumerate((1, 2, 4, 8)):
        maps[f"s1_{l}"] = (fo.kernel_map(c[l], c[l], 3, ts), c[l], c[l])
    for l, ts in enumerate((1, 2, 4)):
        maps[f"s2_{l}"] = (fo.kernel_map(c[l], c[l + 1], 3, ts), c[l], c[l + 1])
    for l, ts in ((3, 4), (2, 2), (1, 1)):
        maps[f"tr_{l}"] = (fo.kernel_map(c[l], c[l - 1], 3, ts, transpose=True), c[l], c[l - 1])
    return maps


def test_oracle_maps_equal_minkowski_engine_cpu_manager():
    """The coordinate maps and the kernel map of EVERY convolution of ResUNetBN2C - conv1 (k = 7), the 3^3 stride-1 maps of the four
    levels, the three stride-2 convolutions and the three TRANSPOSED convolutions - as the reference's own CPU coordinate manager
    builds them (tests/golden/me_maps.npz from oracle/gen_golden_me.py: MinkowskiEngine's src/coordinate_map_manager.cpp compiled
    where it lies, driven exactly as src/convolution_cpu.cpp / convolution_transpose_cpu.cpp drive it).  Compared as sets of (kernel
    index, input coordinate, output coordinate) triples: the reference numbers the rows of a strided map in hash-table order, which
    no result depends on; level 0 is compared row by row (first occurrence, input order)."""
    import hashlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "me_maps.npz"))
    for name in ("surface", "random", "line"):
        q = g[f"{name}_input"]
        sel, c = _oracle_levels(q)
        assert np.array_equal(sel, g[f"{name}_unique_map"])                         # which duplicate survives, in which order
        assert np.array_equal(c[0], g[f"{name}_coords_0"])                          # level-0 rows = input order of the survivors
        for l in (1, 2, 3):
            ref = g[f"{name}_coords_{l}"]
            assert len(ref) == len(c[l]) and set(map(tuple, ref.tolist())) == set(map(tuple, c[l].tolist())), (name, l)
        for mn, (rows, cin, cout) in _oracle_maps(c).items():
            t = _oracle_triples(rows, cin, cout)
            assert len(t) == int(g[f"{name}_{mn}_count"]), (name, mn, len(t), int(g[f"{name}_{mn}_count"]))
            assert hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest() == str(g[f"{name}_{mn}_sha256"]), (name, mn)
            if f"{name}_{mn}_triples" in g.files:
                assert np.array_equal(t, g[f"{name}_{mn}_triples"]), (name, mn)
    # the figure of MinkowskiEngine/tests/python/kernel_map.py: the reference's manager yields 26 pairs (13 per batch item) and the five
    # output coordinates the oracle derives; the 16 that python test asserts is not what the reference's code computes
    pts = np.concatenate([g["figure_points"], np.zeros((8, 1), np.int32)], 1)
    out = fo.stride_coords(pts, 2)
    assert int(g["figure_pairs_total"]) == 26 == 2 * int((fo.kernel_map(pts, out, 3, 1) >= 0).sum())
    assert sorted(map(tuple, out[:, :2].tolist())) == sorted(map(tuple, g["figure_out_coords_item0"].tolist()))


def test_oracle_maps_against_live_minkowski_engine_manager_if_built():
    """the same comparison on a fresh random cloud when oracle/_ref/me_maps.so exists (build container, or shipped to the GPU box)"""
    import pytest
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import build_me_ref
    me = build_me_ref.load()
    if me is None:
        pytest.skip("oracle/_ref/me_maps.so not built here (python oracle/build_me_ref.py needs /root/reference)")
    rs = np.random.RandomState(123)
    q = np.concatenate([rs.randint(-12, 12, size=(700, 3)), rs.randint(-3, 3, size=(200, 3))]).astype(np.int32)
    o = me.fcgf_maps(torch.from_numpy(np.ascontiguousarray(np.concatenate([np.zeros((len(q), 1), np.int32), q], 1))), 7)
    sel, c = _oracle_levels(q)
    assert np.array_equal(sel, o["unique_map"].numpy())
    mc = [o[f"coords_{l}"].numpy()[:, 1:].astype(np.int32) for l in range(4)]
    assert np.array_equal(mc[0], c[0])

    def me_triples(km, cin, cout):
        rows = [np.concatenate([np.full((len(a), 1), k, np.int32), cin[a.numpy()], cout[b.numpy()]], 1) for k, (a, b) in enumerate(km) if len(a)]
        t = np.concatenate(rows).astype(np.int32)
        return t[np.lexsort(t.T[::-1])]
    ref = {"conv1": (o["conv1"], mc[0], mc[0])}
    for l in range(4):
        ref[f"s1_{l}"] = (o[f"conv_s1_{l}"], mc[l], mc[l])
    for l in range(3):
        ref[f"s2_{l}"] = (o[f"conv_s2_{l}"], mc[l], mc[l + 1])
    for l in (3, 2, 1):
        assert bool(o[f"tr_out_is_level_{l}"])
        ref[f"tr_{l}"] = (o[f"conv_tr_{l}"], mc[l], mc[l - 1])
    for mn, (rows, cin, cout) in _oracle_maps(c).items():
        assert np.array_equal(_oracle_triples(rows, cin, cout), me_triples(*ref[mn])), mn


def test_cached_backbone_fixture_is_the_oracles_output():
    """tests/golden/fcgf15.npz (what the GPU test of the 15-copy backbone pass compares copy 7 with) is a cached output of THIS oracle
    (oracle/gen_golden_fcgf15.py): recomputed here - voxelisation digests, the 1024 stored rows, every row sum - so the cache cannot go
    stale behind a change of the oracle, the cloud generator or the weight generator."""
    import hashlib
    from yoho_amd import synth, weights as W
    from yoho_amd.tables import default_tables
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fcgf15.npz"))
    fsd = W.synth_state_dict(W.FCGF_SPEC, int(g["fcgf_seed"]))
    pc = synth.surface_cloud(int(g["points"]), seed=1, extent=float(g["extent"]))
    rot = pc @ default_tables().R64[int(g["copy"])].T
    sel, coords = fo.voxelize(rot, float(g["voxel"]))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(np.asarray(sel, np.int64)) == str(g["sel_sha"]) and sha(np.asarray(coords, np.int32)) == str(g["coords_sha"])
    F = np.asarray(fo.extract_features(rot, float(g["voxel"]), fsd)[1], dtype=np.float32)
    assert F.shape[0] == int(g["n"])
    assert np.abs(F[g["rows"]] - g["feat_rows"]).max() < 2e-6                 # torch's CPU kernels sum in thread-dependent order
    assert np.abs(F.astype(np.float64).sum(1) - g["rowsum"]).max() < 32 * 2e-6
