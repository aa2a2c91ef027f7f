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
