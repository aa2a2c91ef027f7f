"""FCGF backbone on the GPU (csrc/sparse.hip through the C ABI) against oracle/fcgf_oracle.py."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import fcgf_oracle as fo  # noqa: E402
from yoho_amd import synth, weights as W  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-5        # measured 4e-7 .. 1.1e-6 (fp16x2 split products, fp32 accumulation)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def fsd():
    return W.synth_state_dict(W.FCGF_SPEC, 3)


@pytest.fixture(scope="module")
def fctx(hip, fsd):
    c = hip.Context()
    c.load_fcgf(fsd)
    return c


def test_voxelize_first_occurrence(fctx):
    pc = synth.surface_cloud(20000, seed=5)
    sel, coords = fctx.fcgf_voxelize(torch.from_numpy(pc).cuda(), 0.025)
    s0, c0 = fo.voxelize(pc, 0.025)
    assert np.array_equal(sel.cpu().numpy(), s0) and np.array_equal(coords.cpu().numpy(), c0)
    # negative coordinates, duplicates only, a single point
    pc2 = np.concatenate([pc[:50] - 3.0, pc[:50] - 3.0])
    sel, coords = fctx.fcgf_voxelize(torch.from_numpy(pc2).cuda(), 0.025)
    s0, c0 = fo.voxelize(pc2, 0.025)
    assert np.array_equal(sel.cpu().numpy(), s0) and np.array_equal(coords.cpu().numpy(), c0)
    sel, coords = fctx.fcgf_voxelize(torch.from_numpy(pc[:1].copy()).cuda(), 0.025)
    assert sel.tolist() == [0]


def test_voxelize_rotated_equals_voxelising_the_rotated_copy(fctx, tables):
    """yoho_fcgf_voxelize_rotated / yoho_rotate_select: same voxels, same first points and the same down-sampled cloud as
    rotating on the host (pc @ R^T, f64) and voxelising the copy."""
    pc = synth.surface_cloud(30000, seed=9, extent=2.0)
    pc_d = torch.from_numpy(pc).cuda()
    for g in (0, 7, 41):
        R = tables.R64[g]
        rot = pc @ R.T
        s0, c0 = fo.voxelize(rot, 0.025)
        sel, coords, ps = fctx.fcgf_voxelize_rotated(pc_d, R, 0.025)
        assert np.array_equal(sel.cpu().numpy(), s0) and np.array_equal(coords.cpu().numpy(), c0)
        assert np.allclose(ps.cpu().numpy(), rot[s0].astype(np.float32), rtol=0, atol=5e-7)
        kidx = torch.arange(0, 30000, 7, device="cuda")
        q = fctx.rotate_select(pc_d, R, kidx).cpu().numpy()
        assert np.allclose(q, rot[::7].astype(np.float32), rtol=0, atol=5e-7)
    assert np.array_equal(fctx.rotate_select(pc_d, None, kidx).cpu().numpy(), pc[::7].astype(np.float32))


def test_voxelize_rotated_batch_equals_single_calls(fctx, tables):
    """yoho_fcgf_voxelize_rotated_batch (one read-back for all copies) against one call per rotation"""
    pc_d = torch.from_numpy(synth.surface_cloud(7000, seed=12)).cuda()
    Rs = [tables.R64[g] for g in (0, 7, 33, 59)]
    bat = fctx.fcgf_voxelize_rotated_batch(pc_d, Rs, 0.025)
    for R, (sel, coords, ps) in zip(Rs, bat):
        s1, c1, p1 = fctx.fcgf_voxelize_rotated(pc_d, R, 0.025)
        assert torch.equal(sel, s1) and torch.equal(coords, c1) and torch.equal(ps, p1)
    sel, coords = fctx.fcgf_voxelize_rotated_batch(pc_d, Rs[:1], 0.025, want_points=False)[0]
    assert torch.equal(sel, bat[0][0]) and torch.equal(coords, bat[0][1])
    assert fctx.fcgf_voxelize_rotated_batch(pc_d[:0], Rs, 0.025)[2][0].shape[0] == 0      # empty cloud


@pytest.mark.parametrize("n,seed", [(1500, 1), (6000, 2)])
def test_backbone_vs_oracle(fctx, fsd, n, seed):
    pc = synth.surface_cloud(n, seed=seed)
    _, coords = fo.voxelize(pc, 0.025)
    F0 = fo.extract_features(pc, 0.025, fsd)[1]
    F = fctx.fcgf_forward(torch.from_numpy(coords).cuda()).cpu().numpy()
    assert F.shape == F0.shape and np.isfinite(F).all()
    print("fcgf backbone n=%d voxels=%d: rel err %.3g" % (n, len(coords), rel(F, F0)))
    assert rel(F, F0) < TOL
    assert np.allclose(np.linalg.norm(F, axis=1), 1, atol=1e-5)


def test_backbone_row_order_and_translation(fctx):
    pc = synth.surface_cloud(4000, seed=7)
    _, coords = fo.voxelize(pc, 0.025)
    c = torch.from_numpy(coords).cuda()
    F = fctx.fcgf_forward(c)
    perm = torch.randperm(len(coords), generator=torch.Generator().manual_seed(0)).cuda()
    Fp = fctx.fcgf_forward(c[perm].contiguous())
    assert (Fp - F[perm]).abs().max().item() < 2e-5        # rows follow the input order, values do not depend on it
    shift = torch.tensor([16, -8, 24], dtype=torch.int32, device="cuda")
    Fs = fctx.fcgf_forward((c + shift).contiguous())
    assert (Fs - F).abs().max().item() < 2e-5              # multiples of the coarsest stride: all maps shift together


def test_batched_clouds_equal_separate_passes(fctx):
    clouds = [torch.from_numpy(fo.voxelize(synth.surface_cloud(n, seed=sd), 0.025)[1]).cuda() for n, sd in ((3000, 1), (500, 2), (4500, 3))]
    sep = [fctx.fcgf_forward(c) for c in clouds]
    bat = fctx.fcgf_forward_batch(clouds)
    for a, b in zip(sep, bat):
        assert a.shape == b.shape and (a - b).abs().max().item() < 2e-5
    # identical clouds in one batch must not see each other (cloud index is part of the voxel key)
    two = fctx.fcgf_forward_batch([clouds[0], clouds[0]])
    assert (two[0] - sep[0]).abs().max().item() < 2e-5 and torch.equal(two[0], two[1])


def test_internal_row_orders_are_bit_identical(fctx):
    """cell-sorted level-0 rows, parity-sorted rows and skipped kernel offsets of the transposed convolutions change no bit
    of the output, and rows come back in the caller's order"""
    clouds = [torch.from_numpy(fo.voxelize(synth.surface_cloud(n, seed=sd), 0.025)[1]).cuda() for n, sd in ((6000, 5), (37, 6), (2500, 7))]
    clouds.append(torch.from_numpy(fo.voxelize(synth.surface_cloud(20000, seed=8, extent=6.0), 0.025)[1]).cuda())    # wider than the 128-voxel cell wrap
    try:
        fctx.set_fcgf_sort(False, 4)                          # hash-table coordinate maps, first-occurrence rows, no sorting at all
        ref = [fctx.fcgf_forward(c) for c in clouds] + list(fctx.fcgf_forward_batch(clouds))
        outs = []
        # hash-table path with its sorts; then the default path (rank-ordered bitmaps: rows of every level in brick order)
        for par, cells in ((True, 4), (False, 6), (True, 6), (False, 0), (True, 1)):
            fctx.set_fcgf_sort(par, cells)
            outs.append([fctx.fcgf_forward(c) for c in clouds] + list(fctx.fcgf_forward_batch(clouds)))
    finally:
        fctx.set_fcgf_sort(True, 1)                           # the defaults
    for got in outs:
        for a, b in zip(ref, got):
            assert torch.equal(a, b)


def test_duplicate_voxel_rows_are_computed_like_their_first_occurrence(fctx):
    """The C ABI takes 'the distinct voxels' of a cloud; rows that repeat a voxel are still answered - as MinkowskiEngine answers a
    duplicate's lookup with the first row (tests/cpp/coordinate_map_cpu_test.py:47-65): the rank-ordered bitmaps notice the repeat
    (fewer set bits than rows) and hand over to the hash tables, and the maps are then built by every row's own probes (the
    mirrored / inverted maps only ever reach a voxel's first row).  Both coordinate-map paths, same bits."""
    c = torch.from_numpy(fo.voxelize(synth.surface_cloud(5000, seed=4), 0.025)[1]).cuda()
    dup = torch.cat([c[:300], c, c[100:150]])
    try:
        outs = []
        for cells in (1, 5):
            fctx.set_fcgf_sort(True, cells)
            F = fctx.fcgf_forward(c)
            Fd = fctx.fcgf_forward(dup)
            assert torch.equal(Fd[300:300 + c.shape[0]], F) and torch.equal(Fd[:300], F[:300]) and torch.equal(Fd[300 + c.shape[0]:], F[100:150])
            outs.append(Fd)
        assert torch.equal(outs[0], outs[1])
    finally:
        fctx.set_fcgf_sort(True, 1)


def test_other_model_configs_and_tiny_clouds(hip):
    """ResUNetBN2B channels, conv1 kernel 5, normalize_feature off; clouds of 1 and 2 voxels, negative coordinates"""
    spec = W.fcgf_spec((None, 32, 64, 128, 256), (None, 64, 64, 64, 64), 32, 5)
    sd = W.synth_state_dict(spec, 9)
    c = hip.Context()
    c.load_fcgf(sd, channels=(0, 32, 64, 128, 256), tr_channels=(0, 64, 64, 64, 64), conv1_kernel_size=5, normalize_feature=False)
    pc = synth.surface_cloud(2500, seed=4) - 1.7                        # negative voxel indices
    _, coords = fo.voxelize(pc, 0.025)
    F0 = fo.resunet_forward(coords, sd, conv1_kernel_size=5, normalize_feature=False)
    F0 = F0 / np.linalg.norm(F0, axis=1, keepdims=True)                 # fcgf_feat.py:48
    F = c.fcgf_forward(torch.from_numpy(coords).cuda()).cpu().numpy()
    assert rel(F, F0) < TOL
    for cc in (np.array([[3, -4, 5]], np.int32), np.array([[0, 0, 0], [1, 0, 0]], np.int32)):
        F0 = fo.resunet_forward(cc, sd, conv1_kernel_size=5, normalize_feature=False)
        F0 = F0 / np.linalg.norm(F0, axis=1, keepdims=True)
        F = c.fcgf_forward(torch.from_numpy(cc).cuda()).cpu().numpy()
        assert rel(F, F0) < TOL
    assert tuple(c.fcgf_forward(torch.empty((0, 3), dtype=torch.int32, device="cuda")).shape) == (0, 32)
    with pytest.raises(RuntimeError):
        hip.Context().fcgf_forward(torch.zeros((4, 3), dtype=torch.int32, device="cuda"))      # weights not loaded


def test_fp16x2_kernels_against_the_fp32_mfma_variant(hip, fsd, monkeypatch):
    """YOHO_FCGF=f32 (fp32-MFMA sparse convolutions, exact per-bit first convolution; read when the weights are loaded) against
    the default fp16x2 kernels with the first convolution as an MFMA product over the occupancy bits"""
    coords = torch.from_numpy(fo.voxelize(synth.surface_cloud(5000, seed=11), 0.025)[1]).cuda()
    monkeypatch.setenv("YOHO_FCGF", "f32")
    c32 = hip.Context()
    c32.load_fcgf(fsd)
    monkeypatch.delenv("YOHO_FCGF")
    c16 = hip.Context()
    c16.load_fcgf(fsd)
    F32, F16 = c32.fcgf_forward(coords), c16.fcgf_forward(coords)
    assert not torch.equal(F32, F16)                                  # two different arithmetic paths ...
    assert rel(F16.cpu().numpy(), F32.cpu().numpy()) < TOL            # ... that agree to fp32 level


def test_reloading_weights_keeps_every_network_intact(hip, fsd, tables):
    """regression: a second yoho_load_fcgf released a buffer it did not own; later loads then overwrote live weights"""
    c = hip.Context()
    pc = synth.surface_cloud(1500, seed=1)
    _, coords = fo.voxelize(pc, 0.025)
    F0 = fo.extract_features(pc, 0.025, fsd)[1]
    cd = torch.from_numpy(coords).cuda()
    c.load_fcgf(fsd)
    c.load_fcgf(fsd)
    c.load_partI(W.synth_state_dict(W.PARTI_SPEC, 7))
    c.load_partII(W.synth_state_dict(W.PARTII_SPEC, 8))
    assert rel(c.fcgf_forward(cd).cpu().numpy(), F0) < TOL
    # the data-gradient path (tap inversion table) still works after the reloads
    w = torch.randn(64, 32, 1, 13, device="cuda")
    dy = torch.randn(3, 64, 60, device="cuda")
    nei = torch.from_numpy(tables.N.astype(np.int64).reshape(-1)).cuda()
    x = torch.zeros(3, 32, 60, device="cuda", requires_grad=True)
    torch.nn.functional.conv2d(x[:, :, nei].reshape(3, 32, 60, 13), w)[:, :, :, 0].backward(dy)
    assert rel(c.gconv_layer(dy, w, None, transpose=True).cpu().numpy(), x.grad.cpu().numpy()) < 1e-5
    c2 = hip.Context()                                                   # a second context in the same process
    c2.load_fcgf(fsd)
    assert rel(c2.fcgf_forward(cd).cpu().numpy(), F0) < TOL


def test_voxel_indices_outside_the_key_range_are_refused(hip, fsd):
    """the packed 64-bit voxel keys hold 19 bits per axis: far-away or non-finite points must be an error, never an
    aliased (silently wrong) coordinate map"""
    c = hip.Context()
    c.load_fcgf(fsd)
    pc = synth.surface_cloud(800, seed=2)
    for bad in (1e9, -7000.0, np.nan, np.inf):
        p = pc.copy()
        p[17, 1] = bad                                       # -7000 m / 0.025 = -280000 < -2^18
        with pytest.raises(hip.YohoError, match="voxel index"):
            c.fcgf_voxelize(torch.from_numpy(p).cuda(), 0.025)
    sel, coords = c.fcgf_voxelize(torch.from_numpy(pc).cuda(), 0.025)      # the context still works afterwards
    assert sel.shape[0] > 0
    far = coords.clone()
    far[5, 0] = 300000
    with pytest.raises(hip.YohoError, match="voxel index"):
        c.fcgf_forward(far)
    assert c.fcgf_forward(coords).shape[0] == coords.shape[0]


# ---- the backbone at the size it is benchmarked at (bench.py's `fcgf` leg, tools/bench_fcgf.py: 300 k points, extent 3 m, voxel 0.025,
# the 15-copy pass yoho_extractor.run makes four of per fragment) ----------------------------------------------------------------------
def _full_size_pass(fx, pc_d, Rs):
    res = fx.extract_rotated_batch(pc_d, Rs, 0.025)
    return res


def test_backbone_fifteen_copy_pass_at_benchmarked_size_vs_oracle(hip, fsd, tables):
    """What only exists at 300 k points x 15 copies (1.3 M voxels in one pass): runs and 2048-row workgroups that straddle cloud
    boundaries (copy sizes are no multiples of anything), 15 clouds sharing every launch, rank-ordered bitmaps of 15 boxes, the
    1024-entry scan blocks of a level with > 1 M rows.  Copy 7 of the first pass of yoho_extractor.run is compared with the
    oracle ROW FOR ROW (selected points and voxel coordinates bit-exact, features to 1e-5), copies 0 and 14 with the oracle's voxelisation
    and with a pass of that copy alone (1e-5), through the same calls the extractor makes
    (fcgf_extractor.extract_rotated_batch = yoho_fcgf_voxelize_rotated_batch -> yoho_fcgf_forward_batch); the hash-table coordinate maps
    give the same bits as the default bitmaps at this size, and five repeats of the pass are bit-identical."""
    from yoho_amd.fcgf_feat import fcgf_extractor
    pc = synth.surface_cloud(300000, seed=1, extent=3.0)
    ctx = hip.Context()
    ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7},
          "state_dict": {k: torch.from_numpy(np.array(v)) for k, v in fsd.items()}}
    fx = fcgf_extractor(ck, ctx=ctx)
    pc_d = torch.from_numpy(pc).cuda()
    Rs = [tables.R64[g] for g in range(15)]
    res = _full_size_pass(fx, pc_d, Rs)
    sizes = [int(sel.shape[0]) for sel, _, _ in res]
    total = sum(sizes)
    assert total > 1_000_000 and len(set(sizes)) > 1
    # cloud boundaries inside the 2048-row workgroups / 1024-row runs of the internal sorts and reductions
    bounds = np.cumsum(sizes)[:-1]
    assert all(b % 2048 != 0 for b in bounds) and all(b % 1024 != 0 for b in bounds)
    worst = 0.0
    # The oracle pins an INTERIOR copy row for row (its output for that copy is a cached fixture); the first and the last copy - whose
    # rows sit at the two ends of every shared launch - are pinned to it through the library itself: their voxelisation against the
    # oracle's, their features against a pass of that copy ALONE, which the oracle pins at 1.5 k - 6 k points (test_backbone_vs_oracle).
    # Not bit for bit at this size: a pass picks the first convolution's kernel by what fits its workspace (occupancy-bitmap MFMA kernel or
    # hash probes; the 7^3 sums then differ in the last bit), and a 1.3 M-row pass and an 87 k-row pass pick differently; at equal kernel
    # choice a cloud's rows do not depend on which clouds share its pass (test_batched_clouds_equal_separate_passes).
    ends = []
    for j in (0, 7, 14):
        rot = pc @ Rs[j].T
        s0, c0 = fo.voxelize(rot, 0.025)
        sel, F, ps = res[j]
        assert np.array_equal(sel.cpu().numpy(), s0), j
        assert np.allclose(ps.cpu().numpy(), rot[s0].astype(np.float32), rtol=0, atol=5e-7)
        if j != 7:
            alone = fx.extract_rotated_batch(pc_d, [Rs[j]], 0.025)[0]
            assert torch.equal(alone[0], sel) and torch.equal(alone[2], ps), j
            d_alone = float((alone[1] - F).abs().max())
            ends.append(d_alone)
            assert d_alone < TOL, (j, d_alone)                        # unit rows: absolute = relative
            continue
        # copy 7 against the oracle's output for it, cached by oracle/gen_golden_fcgf15.py (the same oracle call this test used to make
        # on the GPU box's host for ~75 s per run): 1024 full rows and the row sum of every row
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fcgf15.npz"))
        assert int(g["copy"]) == 7 and int(g["n"]) == len(s0)
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        assert sha(np.asarray(s0, np.int64)) == str(g["sel_sha"]) and sha(np.asarray(c0, np.int32)) == str(g["coords_sha"])     # the oracle's voxelisation, as cached
        Fj = F.cpu().numpy()
        assert Fj.shape == (len(s0), 32) and np.isfinite(Fj).all()
        e = rel(Fj[g["rows"]], g["feat_rows"])
        worst = max(worst, e)
        assert e < TOL, (j, e)
        assert np.abs(Fj[g["rows"]] - g["feat_rows"]).max(axis=1).max() < 5e-6           # row for row (rows are unit vectors: absolute = relative)
        assert np.abs(Fj.astype(np.float64).sum(1) - g["rowsum"]).max() < 32 * 5e-6      # every row: a wrong or misplaced row changes its sum
    print("fcgf backbone, 15-copy pass, %d voxels (copies %d..%d): worst rel err of copy 7 vs the oracle %.3g; copies 0 / 14 vs their single-cloud passes: max abs diff %s" % (total, min(sizes), max(sizes), worst, ["%.2g" % v for v in ends]))
    # the same pass: five repeats, and once with hash-table coordinate maps - identical bits
    for _ in range(5):
        again = _full_size_pass(fx, pc_d, Rs)
        for (s1, f1, p1), (s2, f2, p2) in zip(res, again):
            assert torch.equal(s1, s2) and torch.equal(f1, f2) and torch.equal(p1, p2)
    try:
        ctx.set_fcgf_sort(True, 4 | 1)
        hashed = _full_size_pass(fx, pc_d, Rs)
    finally:
        ctx.set_fcgf_sort(True, 1)
    for (s1, f1, _), (s2, f2, _) in zip(res, hashed):
        assert torch.equal(s1, s2) and torch.equal(f1, f2)


def test_backbone_wide_and_sparse_clouds_share_a_pass(hip, fsd):
    """A cloud far wider than one brick of the rank-ordered bitmaps / the 128-voxel cell wrap (9 m = 360 voxels per axis), a 37-point
    cloud and a mid-sized one in one pass, against the oracle; and the case ADVICE r4 found: clouds that are SPARSE in LARGE boxes
    (15 x 5 k voxels over 800 x 800 x 240 cells: 660 MB of rank arrays against a 670 MB workspace estimate that only knows the voxel
    count) - the pass must grow its workspace (or fall back to the hash tables), never return YOHO_ENOMEM, and give the hash path's bits."""
    ctx = hip.Context()
    ctx.load_fcgf(fsd)
    pcs = [synth.surface_cloud(30000, seed=21, extent=9.0), synth.surface_cloud(37, seed=22), synth.surface_cloud(9000, seed=23, extent=1.5)]
    vox = [fo.voxelize(p, 0.025) for p in pcs]
    clouds = [torch.from_numpy(c).cuda() for _, c in vox]
    assert (vox[0][1].max(0) - vox[0][1].min(0)).max() > 320
    outs = ctx.fcgf_forward_batch(clouds)
    for p, F in zip(pcs, outs):
        F0 = fo.extract_features(p, 0.025, fsd)[1]
        assert rel(F.cpu().numpy(), F0) < TOL
    # sparse clouds in large boxes, fresh context (small workspace)
    rs = np.random.RandomState(5)
    sparse = []
    for b in range(15):
        c = np.unique(np.stack([rs.randint(0, 800, 5200), rs.randint(0, 800, 5200), rs.randint(0, 240, 5200)], 1).astype(np.int32), axis=0)
        c = c[rs.permutation(len(c))] + np.array([-400 + 7 * b, -123, 50 * b], np.int32)
        sparse.append(torch.from_numpy(np.ascontiguousarray(c)).cuda())
    fresh = hip.Context()
    fresh.load_fcgf(fsd)
    got = fresh.fcgf_forward_batch(sparse)
    again = fresh.fcgf_forward_batch(sparse)                  # second pass of the same shape: the grown workspace is kept
    hashed_ctx = hip.Context()
    hashed_ctx.load_fcgf(fsd)
    hashed_ctx.set_fcgf_sort(True, 4 | 1)
    hashed = hashed_ctx.fcgf_forward_batch(sparse)
    for a, b, c in zip(got, again, hashed):
        assert torch.equal(a, b) and torch.equal(a, c)
    F0 = fo.resunet_forward(sparse[3].cpu().numpy(), fsd)
    F0 = F0 / np.linalg.norm(F0, axis=1, keepdims=True)
    assert rel(got[3].cpu().numpy(), F0) < TOL


def test_enomem_recoveries_run_and_leave_no_error_behind(hip, fsd, tables, monkeypatch):
    """both YOHO_ENOMEM recoveries of the backbone actually run (YOHO_WS_LIMIT_MB makes a workspace request above the limit
    fail exactly as hipMalloc on an exhausted device does) - (a) the pass whose bitmaps do not fit: first attempt -> grown workspace
    refused -> third attempt on the hash tables; (b) the batched voxelisation whose rank arrays cannot be had -> the table path.  Each
    must succeed (no stale HIP error surfacing behind the next launch as YOHO_EHIP), leave yoho_last_error empty and give the unlimited
    context's result: (b) bit for bit (integers); (a) to the backbone's tolerance - the last-resort attempt cannot grow the workspace for
    the first convolution's occupancy bitmaps either, so that layer runs on its hash-probe kernel, whose sums differ from the bitmap
    kernel's in the last bit (measured 1.2e-7 - 1.8e-7 absolute on unit rows)."""
    rs = np.random.RandomState(5)
    sparse = []
    for b in range(15):
        c = np.unique(np.stack([rs.randint(0, 800, 5200), rs.randint(0, 800, 5200), rs.randint(0, 240, 5200)], 1).astype(np.int32), axis=0)
        c = c[rs.permutation(len(c))] + np.array([-400 + 7 * b, -123, 50 * b], np.int32)
        sparse.append(torch.from_numpy(np.ascontiguousarray(c)).cuda())
    free = hip.Context()
    free.load_fcgf(fsd)
    want = free.fcgf_forward_batch(sparse)
    # 360 voxels per axis (up to ~620 for a rotated copy, still inside one bitmap's 2^24 words): 0.2-0.9 GB of bitmaps + rank arrays for 15 copies
    pc_d = torch.from_numpy(synth.surface_cloud(20000, seed=12, extent=9.0)).cuda()
    Rs = [tables.R64[g] for g in range(15)]
    want_v = free.fcgf_voxelize_rotated_batch(pc_d, Rs, 0.025)
    monkeypatch.setenv("YOHO_WS_LIMIT_MB", "1300")       # first attempt asks for 1.13 GB, the grown one for 1.8
    lim = hip.Context()
    monkeypatch.delenv("YOHO_WS_LIMIT_MB")
    lim.load_fcgf(fsd)
    lib = hip.load_library()
    got = lim.fcgf_forward_batch(sparse)
    assert lib.yoho_last_error() == b"", lib.yoho_last_error()
    for a, b in zip(got, want):
        assert rel(a.cpu().numpy(), b.cpu().numpy()) < TOL
    assert any(not torch.equal(a, b) for a, b in zip(got, want)), "the limit did not force the last-resort attempt (its first layer differs in the last bit)"
    monkeypatch.setenv("YOHO_WS_LIMIT_MB", "100")
    lim2 = hip.Context()
    monkeypatch.delenv("YOHO_WS_LIMIT_MB")
    got_v = lim2.fcgf_voxelize_rotated_batch(pc_d, Rs, 0.025)
    assert lib.yoho_last_error() == b"", lib.yoho_last_error()
    for a, b in zip(got_v, want_v):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    # and the limited context still works for an ordinary pass afterwards
    small = [torch.from_numpy(fo.voxelize(synth.surface_cloud(3000, seed=4), 0.025)[1]).cuda()]
    assert torch.equal(lim.fcgf_forward_batch(small)[0], free.fcgf_forward_batch(small)[0])
