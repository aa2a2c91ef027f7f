"""Host-side logic that needs no GPU: batching of backbone passes, nearest-neighbour hint plumbing."""
import types

import numpy as np
import torch

from yoho_amd.fcgf_feat import fcgf_extractor


class _FakeCtx:
    """records what the extractor asks the library for; clouds are (n,3) int tensors, features are their row counts"""

    def __init__(self, sizes):
        self.sizes, self.batches = list(sizes), []
        self._i = 0
        self.fcgf_owner = None
        self.loads = 0

    def load_fcgf(self, owner=None, **kw):
        self.fcgf_owner = owner
        self.loads += 1

    def fcgf_voxelize(self, pts, voxel_size):
        n = self.sizes[self._i]
        self._i += 1
        return torch.arange(n), torch.zeros((n, 3), dtype=torch.int32)

    def fcgf_forward_batch(self, coords_list):
        self.batches.append([c.shape[0] for c in coords_list])
        return [torch.full((c.shape[0], 1), float(len(self.batches))) for c in coords_list]


def _extractor(ctx, budget):
    ex = object.__new__(fcgf_extractor)           # no checkpoint / library needed for the batching logic
    ex.ctx = ctx
    ex.MAX_VOXELS_PER_PASS = budget
    ex._load_args = {}
    return ex


def test_backbone_batches_respect_voxel_budget_and_order():
    sizes = [400, 500, 300, 900, 100, 100, 100, 1200, 50]
    ctx = _FakeCtx(sizes)
    out = _extractor(ctx, 1000).extract_features_dev_batch([None] * len(sizes), 0.025)
    assert ctx.batches == [[400, 500], [300], [900, 100], [100, 100], [1200], [50]]     # a single oversize cloud still goes alone
    assert [f.shape[0] for _, f in out] == sizes                                        # results come back in input order
    assert [int(f[0, 0]) for _, f in out] == [1, 1, 2, 3, 3, 4, 4, 5, 6]


def test_backbone_batches_cap_at_64_clouds():
    ctx = _FakeCtx([10] * 150)
    _extractor(ctx, 10 ** 9).extract_features_dev_batch([None] * 150, 0.025)
    assert [len(b) for b in ctx.batches] == [64, 64, 22]


def test_extractor_pass_sizes_cover_the_group_once():
    """yoho_extractor._pass_starts: YOHO_ROT_BATCH as one number or a list of pass sizes - every group element in exactly one pass"""
    from yoho_amd.yoho_extract import yoho_extractor
    ex = object.__new__(yoho_extractor)
    for rb, want in ((15, [0, 15, 30, 45, 60]), (16, [0, 16, 32, 48, 60]), (60, [0, 60]), (100, [0, 60]), ([9, 17, 17, 17], [0, 9, 26, 43, 60]),
                     ([30], [0, 30, 60]), ([7, 20], [0, 7, 27, 47, 60]), ([0, 59], [0, 1, 60]), ([70, 5], [0, 60])):
        ex.rot_batch = rb
        assert ex._pass_starts(60) == want, rb


def test_lane_context_loads_the_backbone_once_per_context(monkeypatch):
    """fcgf_extractor.lane_context(): the second lane's context is process-wide (hip.get_context(lane=1)) and gets this backbone's
    weights when another object's are resident; a pass on it leaves the first context's residency alone"""
    from yoho_amd import hip
    main, lane = _FakeCtx([]), _FakeCtx([])
    main.device, main.tables = 0, types.SimpleNamespace(dir="/nowhere")
    asked = []
    monkeypatch.setattr(hip, "get_context", lambda device=None, so3_dir=None, lane=0: (asked.append((device, so3_dir, lane)), lane_ctx)[1])
    lane_ctx = lane
    ex = _extractor(main, 1000)
    main.fcgf_owner = ex
    assert ex.lane_context() is lane and asked == [(0, "/nowhere", 1)] and lane.loads == 1 and lane.fcgf_owner is ex
    assert ex.lane_context() is lane and lane.loads == 1 and len(asked) == 1          # kept, resident
    lane.fcgf_owner = object()                                                         # another backbone used the shared lane context
    ex.lane_context()
    assert lane.loads == 2 and lane.fcgf_owner is ex and main.loads == 0


def test_gather_sets_and_clears_the_grid_hint():
    from yoho_amd import gather
    calls = []
    ctx = types.SimpleNamespace(set_nn_grid=lambda c: calls.append(("grid", c)),
                                group_gather=lambda k, p, f, g, out: calls.append(("g", g)))
    orig = torch.Tensor.to
    try:
        torch.Tensor.to = lambda self, *a, **k: self                      # no device here
        out = torch.zeros((4, 32, 60))
        gather.gather_group_features(np.zeros((4, 3)), [np.zeros((5, 3), np.float32)] * 60, [np.zeros((5, 32), np.float32)] * 60,
                                     ctx=ctx, out=out, voxel_size=0.025)
    finally:
        torch.Tensor.to = orig
    assert calls[0] == ("grid", 0.025) and calls[-1] == ("grid", 0) and [c for c in calls if c[0] == "g"] == [("g", g) for g in range(60)]


def test_power_monitor_degrades_without_smu_access():
    """bench.py's clock / power section must never break the bench line: without a GPU / SMU the sampler reports None fields"""
    from yoho_amd.power import PowerMonitor
    m = PowerMonitor(0)
    one = m.read_once()
    assert set(one) == {"sclk_mhz", "power_w"}
    m.start()
    out = m.stop()
    assert set(out) >= {"sclk_mhz_mean", "power_w_mean", "power_cap_w", "samples", "source"}
    if not m.available:
        assert one == {"sclk_mhz": None, "power_w": None} and out["samples"] == 0


def test_bench_dataset_scene_builder_and_presets(tmp_path):
    """tools/bench_dataset.py: the synthetic scene lands in the reference's on-disk layout (gt.log / gt.info / keypoints / group
    feature files) with exactly the requested pair count, and the 3DMatch preset has that test set's fragment and pair counts"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_dataset as bd
    from yoho_amd.dataset import ThrDMatchPartDataset
    root, cache = str(tmp_path / "origin" / "s"), str(tmp_path / "cache" / "s")
    pairs = bd.build_scene(root, cache, 7, 40, 3, device="cpu", npairs=11)
    ds = ThrDMatchPartDataset(root, 7)
    assert len(pairs) == 11 and [tuple(int(v) for v in p) for p in ds.pair_ids] == pairs
    x = np.load(f"{cache}/FCGF_Input_Group_feature/3.npy")
    assert x.shape == (40, 32, 60) and x.dtype == np.float32 and np.allclose(np.linalg.norm(x, axis=1), 1, atol=1e-5)
    assert ds.get_kps("3").shape == (40, 3)
    assert sum(n for _, n, _ in bd.PRESET_3DMATCH) == 433 and sum(p for _, _, p in bd.PRESET_3DMATCH) == 1623


def test_library_vote_order_is_numpys_shuffle():
    """yoho_vote_order (host code of the library, no device): numpy's RandomState(seed).shuffle(arange(M)) - MT19937 seeded by
    init_genrand, Fisher-Yates from the top with masked rejection sampling - for seeds at both ends of the 32-bit range and sizes
    around the power-of-two mask boundaries; this is what makes yoho_register_pair's YOHO-O vote the reference's
    (tests/estimator.py:321-323) for a seeded generator."""
    from yoho_amd import hip
    for seed in (0, 1, 5, 123456789, 2 ** 32 - 1, 3141592653):
        for M in (0, 1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 625, 1000, 1023, 1024, 1025, 3233, 5000, 16385):
            a = np.arange(M)
            np.random.RandomState(seed).shuffle(a)
            assert np.array_equal(a, hip.vote_order(seed, M)), (seed, M)
    assert np.array_equal(hip.vote_order(2 ** 32 + 5, 100), hip.vote_order(5, 100))       # the seed is taken modulo 2^32


def _yohoc_host(inlier=0.07):
    """a yohoc instance without a library context (the draw / statistic / mask methods are host code)"""
    from yoho_amd.estimator import yohoc
    est = object.__new__(yohoc)
    est.inliner_dist, est.lapack_parity, est.device_sampling = inlier, True, False
    return est


def test_library_yohoc_draws_are_numpys_choice_calls():
    """yoho_c_draw_np (host code of the library): the reference's sampling loop (tests/estimator.py:113-128) taken from numpy's
    legacy stream in C - RandomState.choice(range(60), p) = one 53-bit random_sample + searchsorted(cdf, side='right'),
    choice(bucket, 3) = masked-rejection randint - against np.random.choice itself, call by call: seeds at both ends of the 32-bit
    range, a stream position in the middle of the state block, bucket sizes around the power-of-two mask boundaries, buckets of
    fewer than two matches (zero weight), few buckets, and the state AFTER the draws (the next pair continues the stream)."""
    est = _yohoc_host()
    cases = []
    rs = np.random.RandomState(99)
    for M, nrot in ((3233, 60), (700, 60), (64, 7), (5, 2), (40, 60), (2, 1), (1025 * 3, 3)):
        dr = rs.randint(0, nrot, size=M)
        if nrot == 60 and M > 100:
            dr[rs.rand(M) < 0.5] = 42                       # a dominant coarse rotation, as a registrable pair has
        cases.append(dr)
    for seed in (0, 1, 1234, 2 ** 32 - 1, 3141592653):
        for burn in (0, 311):
            for dr in cases:
                for it in (1, 50, 1000):
                    buckets, prob = est.DR_statictic(dr)
                    if prob is None:
                        continue
                    np.random.seed(seed)
                    np.random.random_sample(burn)
                    want = est._draw_numpy(buckets, prob, it)
                    after = np.random.get_state()
                    np.random.seed(seed)
                    np.random.random_sample(burn)
                    prob2, table = est._statistic_flat(dr)
                    assert np.array_equal(prob, prob2)
                    got = est._draw(None, prob2, it, table=table)
                    mine = np.random.get_state()
                    assert got.dtype == np.int64 and np.array_equal(got, want), (seed, burn, len(dr), it)
                    assert np.array_equal(after[1], mine[1]) and after[2] == mine[2], (seed, burn, len(dr), it)
    # the list form (DR_statictic's dict of lists) gives the same table
    dr = cases[0]
    buckets, prob = est.DR_statictic(dr)
    np.random.seed(5)
    a = est._draw(buckets, prob, 100)
    np.random.seed(5)
    b = est._draw(None, prob, 100, table=est.bucket_table(dr))
    assert np.array_equal(a, b)


def test_library_yohoc_draws_hit_the_draw_limit_like_the_reference():
    """nearly all the weight on rotations that cannot be sampled cannot happen (their weight is 0), so the 50001-draw limit is
    reached only through max_iter; what can happen is that the loop ends by max_iter with `draws == max_iter` - and a statistic
    with no bucket of two matches returns None before any draw (the reference's recalltime 50001)."""
    from yoho_amd import hip
    est = _yohoc_host()
    dr = np.arange(60)                                       # every bucket holds one match
    assert est._statistic_flat(dr)[0] is None and est.DR_statictic(dr) == (None, None)
    assert est._statistic_flat(np.array([3, 3, 7, 9, 9, 9]))[0] is None           # total weight 6e-6 < 1e-4 (:47-48)
    dr = np.array([3] * 30 + [7] + [9] * 40)
    prob, (start, members) = est._statistic_flat(dr)
    st = np.random.RandomState(1)
    tri, draws = hip.c_draw_np(prob, start, members, 200, rng=st)
    assert draws == 200 and tri.shape == (200, 3) and 30 not in set(np.unique(tri))
    # a private RandomState is advanced, the global one is not touched
    g = np.random.get_state()
    hip.c_draw_np(prob, start, members, 10, rng=st)
    assert np.array_equal(g[1], np.random.get_state()[1]) and g[2] == np.random.get_state()[2]


def test_reflection_mask_vectorised_equals_the_reference_expression_per_triple():
    """estimator.yohoc._reflect_mask forms the 3x3 covariances of all sampled triples with stacked numpy calls; the sign LAPACK
    gives the null direction hangs on their last bits, so the stack must carry the bits of the reference's expression on ONE
    triple (tests/estimator.py:56-58): checked on 20000 triples incl. repeated points (rank-1 / rank-0 covariances)."""
    from yoho_amd.estimator import yohoc
    rs = np.random.RandomState(3)
    k0 = rs.rand(3000, 3) * 3.0
    k1 = k0 @ np.linalg.qr(rs.randn(3, 3))[0].T + 0.01 * rs.randn(3000, 3)
    tri = rs.randint(0, 3000, size=(20000, 3))
    tri[:500, 1] = tri[:500, 0]
    tri[500:600, 2] = tri[500:600, 1] = tri[500:600, 0]
    a, b = k0[tri], k1[tri]
    c0 = a - np.mean(a, axis=1, keepdims=True)
    c1 = b - np.mean(b, axis=1, keepdims=True)
    cov = np.matmul(np.swapaxes(c1, 1, 2), c0)
    for i in range(0, 20000, 7):
        one = (b[i] - np.mean(b[i], 0, keepdims=True)).T @ (a[i] - np.mean(a[i], 0, keepdims=True))
        assert np.array_equal(one, cov[i]), i
    assert np.array_equal(yohoc._reflect_mask(a, b), yohoc._reflect_mask_one_by_one(a, b))
