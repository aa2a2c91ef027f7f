"""CPU tests: the oracle (oracle/yoho_oracle.py) against the golden vectors captured from the
real reference by oracle/gen_golden.py.  This is what pins the oracle."""
import os
import sys
import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import yoho_oracle as orc  # noqa: E402
from yoho_amd import synth  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def test_partI_forward_matches_reference(gold, sd1, tables):
    g = gold("partI.npz")
    x = synth.unit_features(16, seed=int(g["xseed"]))
    assert np.array_equal(x, g["x"]), "synthetic input generator is not bit-reproducible"
    eqv, inv = orc.partI_forward(x, sd1, tables.N)
    assert rel(eqv, g["eqv"]) < 2e-6 and rel(inv, g["inv"]) < 2e-6
    eqv_t, inv_t = orc.partI_forward_torch(x, sd1, tables.N)
    assert rel(eqv_t, g["eqv"]) < 2e-6 and rel(inv_t, g["inv"]) < 2e-6


def test_partI_equivariance_property(sd1, tables):
    # SURVEY section 4 property 1: PartI(x[:,:,P[i]]).eqv == PartI(x).eqv[:,:,P[i]]
    x = synth.unit_features(4, seed=2)
    e0, i0 = orc.partI_forward(x, sd1, tables.N)
    for i in (1, 17, 59):
        e, iv = orc.partI_forward(np.ascontiguousarray(x[:, :, tables.P[i]]), sd1, tables.N)
        assert rel(e, e0[:, :, tables.P[i]]) < 1e-5 and rel(iv, i0) < 1e-5


def test_table_identities(tables):
    R, N, P = tables.R64, tables.N, tables.P
    def idx(M):
        d = np.abs(R - M[None]).reshape(60, 9).max(1)
        j = int(np.argmin(d)); assert d[j] < 1e-3
        return j
    assert np.array_equal(N[:, 0], np.arange(60)) and np.array_equal(P[0], np.arange(60))
    for i in range(60):
        assert sorted(P[i]) == list(range(60))
    for i in (0, 5, 33):
        for g in (0, 7, 59):
            assert P[i, g] == idx(R[g] @ R[i])
    for g in (0, 11, 42):
        for k in range(13):
            assert N[g, k] == idx(R[N[0, k]] @ R[g])
    assert np.array_equal(N[P[:, :, None], np.arange(13)[None, None, :]], P[np.arange(60)[:, None, None], N[None, :, :]])
    one, two = tables.cone()
    assert len(one) == 13 and len(two) == 45


def test_pdist_order_bitexact(gold):
    g = gold("pdist.npz")
    assert np.array_equal(orc.pdist_l2(g["A"], g["B"], squared=True), g["d2"])
    d = orc.pdist_l2(g["A"], g["B"])
    ulp = np.abs(d.view(np.int32) - g["dist"].view(np.int32))
    assert ulp.max() <= 1 and (ulp > 0).mean() < 0.02     # torch-CPU (MKL VML) sqrt is not correctly rounded


def test_group_mean_recipe_is_numpy(gold):
    x = synth.unit_features(64, seed=9)
    assert np.array_equal(orc.group_mean_np_explicit(x), np.mean(x, axis=-1))


def _pair(g):
    return synth.make_pair(int(g["K"]), seed=int(g["pair_seed"]))


def test_chain_partI_and_match(gold, sd1, tables):
    g = gold("chain.npz")
    pr = _pair(g)
    eqv0 = orc.partI_extract(pr["feat0"], sd1, tables.N, batch=40)
    eqv1 = orc.partI_extract(pr["feat1"], sd1, tables.N, batch=40)
    assert rel(eqv0[:8], g["eqv0_head"]) < 2e-6 and rel(eqv1[:8], g["eqv1_head"]) < 2e-6
    assert np.allclose(eqv0.astype(np.float64).sum(axis=(1, 2)), g["eqv0_rowsum"], atol=1e-4)
    inv0, inv1 = orc.group_mean_np(eqv0), orc.group_mean_np(eqv1)
    assert rel(inv0, g["inv0"]) < 2e-6
    # matching on the reference's own means must be bit-identical
    m = orc.mutual_match(g["inv0"], g["inv1"])
    assert np.array_equal(m, g["match"]) and m.dtype == np.int64
    # and on our recomputed descriptors (fp32 noise could only flip a near-tie; none here)
    assert np.array_equal(orc.mutual_match(inv0, inv1), g["match"])


def test_chain_des2r(gold, sd1, tables):
    g = gold("chain.npz")
    pr = _pair(g)
    eqv0 = orc.partI_extract(pr["feat0"], sd1, tables.N, batch=40)
    eqv1 = orc.partI_extract(pr["feat1"], sd1, tables.N, batch=40)
    m = g["match"]
    cor = orc.des2r_cor(eqv1[m[:, 1]], eqv0[m[:, 0]], tables.P)
    assert rel(cor, g["cor"]) < 1e-5
    assert np.array_equal(np.argmax(cor, 1), g["dr_index"])
    inl = ~pr["is_out"][m[:, 0]]
    assert (g["dr_index"][inl] == pr["gi"]).mean() > 0.9     # SURVEY section 4 property 2


def test_chain_partII_hyp_yohoo_yohoc(gold, sd1, sd2, tables):
    g = gold("chain.npz")
    pr = _pair(g)
    eqv0 = orc.partI_extract(pr["feat0"], sd1, tables.N, batch=40)
    eqv1 = orc.partI_extract(pr["feat1"], sd1, tables.N, batch=40)
    m, dr = g["match"], g["dr_index"]
    b = orc.batch_create(pr["feat0"][m[:, 0]], pr["feat1"][m[:, 1]], eqv0[m[:, 0]], eqv1[m[:, 1]], dr)
    q = orc.partII_forward(b["before_eqv0"], b["before_eqv1"], b["after_eqv0"], b["after_eqv1"], b["pre_idx"],
                           sd2, tables.N, tables.P)
    assert rel(q[:16], g["quat16"]) < 2e-5
    k0, k1 = pr["keys0"][m[:, 0]], pr["keys1"][m[:, 1]]
    T = orc.hyp_from_quat(q, dr, k0, k1, tables.R32)
    assert rel(T, g["trans_pre"]) < 5e-5
    # estimator on the reference's own hypotheses: exact
    Tref = g["trans_pre"]
    np.random.seed(1234)
    order = np.arange(Tref.shape[0]); np.random.shuffle(order)
    bid, cnt, Tb = orc.yohoo_select(k0, k1, Tref, order, 0.09, 1000)
    assert bid == int(g["yohoo_recall"]) and np.array_equal(Tb, g["yohoo_trans"])
    np.random.seed(4321)
    order = np.arange(Tref.shape[0]); np.random.shuffle(order)
    bid, cnt, Tb = orc.yohoo_select(k0, k1, Tref, order, 0.09, 20)
    assert bid == int(g["yohoo20_recall"]) and np.array_equal(Tb, g["yohoo20_trans"])
    # YOHO-C with the same global RNG stream
    np.random.seed(99)
    tri = orc.yohoc_draw_triples(dr, 200, np.random)
    it, cnt, Tc, dets = orc.yohoc_select(k0, k1, tri, 0.07)
    assert it == int(g["yohoc_recall"]) and np.allclose(Tc, g["yohoc_trans"], atol=1e-12)
    assert 0 < (dets < 0).mean() < 1          # the reference really produces reflections
    text = orc.r_pre_log_text(["0", "1"], [("0", "1")], {("0", "1"): g["yohoo_trans"]})
    assert text == str(g["prelog_o"])


def test_quat_and_kabsch(gold):
    g = gold("quat.npz")
    for q, M in zip(g["q"], g["mats"]):
        assert np.array_equal(orc.matrix_from_quaternion(q), M)
    g = gold("kabsch.npz")
    for k0, k1, T in zip(g["k0"], g["k1"], g["T"]):
        Tr, s = orc.threepps2tran(k0, k1)
        assert np.allclose(Tr, T, atol=1e-12)
        Tp, _ = orc.threepps2tran(k0, k1, proper=True)
        assert abs(np.linalg.det(Tp[:, :3]) - 1) < 1e-9
        # proper and reference variants agree on the three sample points themselves
        assert np.allclose(orc.transform_points(k1, Tp), orc.transform_points(k1, Tr), atol=1e-9)


def test_group_gather_small(tables):
    rs = np.random.RandomState(0)
    keys = rs.rand(40, 3) * 2
    pts = [(rs.rand(300, 3) * 2).astype(np.float32) for _ in range(60)]
    feats = [rs.randn(300, 32).astype(np.float32) for _ in range(60)]
    out = orc.group_gather(keys, pts, feats, tables.R64)
    assert out.shape == (40, 32, 60) and out.dtype == np.float32
    g = 7
    kr = keys @ tables.R64[g].T
    j = np.argmin(((kr[:, None] - pts[g][None].astype(np.float64)) ** 2).sum(-1), 1)
    assert np.array_equal(out[:, :, g], feats[g][j])


def test_philox_known_answers_and_device_sampler_statistic():
    """Philox4x32-10 against the Random123 known-answer vectors; the oracle's restatement of the device-side YOHO-C
    sampler draws from the same statistic as the reference's DR_statictic (tests/estimator.py:34-51)"""
    assert orc.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert orc.philox4x32_10((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert orc.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
    rs = np.random.RandomState(0)
    dr = rs.randint(0, 60, 400)
    dr[:150] = 17
    dr[150:210] = 42
    tri = orc.yohoc_device_triples(dr, 4000, seed=2024)
    assert tri.shape == (4000, 3) and tri.dtype == np.int64
    buckets, prob = orc.dr_statistic(dr)
    rot = dr[tri[:, 0]]
    assert (dr[tri[:, 1]] == rot).all() and (dr[tri[:, 2]] == rot).all()        # the three matches share a coarse rotation
    freq = np.bincount(rot, minlength=60) / 4000.0
    assert np.abs(freq - prob).max() < 0.03                                      # drawn with the reference's weights
    assert (prob[np.bincount(dr, minlength=60) < 2] == 0).all() and (freq[prob == 0] == 0).all()
    assert orc.yohoc_device_triples(np.arange(60), 10, 1) is None                # no bucket with two matches
    assert np.array_equal(tri, orc.yohoc_device_triples(dr, 4000, seed=2024)) and not np.array_equal(tri, orc.yohoc_device_triples(dr, 4000, seed=2025))


def test_device_sampler_follows_the_reference_sampling_law():
    """Statistical parity of YOHO-C's device mode, pinned as a distribution: the library's sampler is bit-exact against
    orc.yohoc_device_triples (tests/test_gpu_dropin.py), and that restatement is tested here against the law the reference draws from
    (tests/estimator.py:119-128: a coarse rotation with probability p_b ~ n (n - .01)(n - .02), then three members of its bucket uniformly
    WITH replacement, independently): chi-square of the bucket counts against p, of the member counts of the largest bucket against
    uniform, of the (first, second) member pairs against the product law, and a two-sample chi-square against draws made by numpy
    exactly as the reference makes them.  Thresholds are the 99.9 % quantiles (Wilson-Hilferty); the seed is fixed, so the test is
    deterministic."""
    def chi2_crit(dof, z=3.09):                                  # upper 0.1 % quantile of chi-square(dof)
        return dof * (1.0 - 2.0 / (9.0 * dof) + z * np.sqrt(2.0 / (9.0 * dof))) ** 3

    rs = np.random.RandomState(5)
    dr = rs.randint(0, 60, 900)
    dr[:200] = 17
    dr[200:290] = 42
    dr[290:330] = 3
    n_it = 20000
    tri = orc.yohoc_device_triples(dr, n_it, seed=0x1234567890ABCDEF)
    buckets, prob = orc.dr_statistic(dr)
    live = np.nonzero(prob > 0)[0]
    rot = dr[tri[:, 0]]
    obs = np.bincount(rot, minlength=60)[live].astype(np.float64)
    exp = prob[live] * n_it
    big = exp >= 5                                               # pool the rare buckets
    o = np.append(obs[big], obs[~big].sum())
    e = np.append(exp[big], exp[~big].sum())
    keep = e > 0
    stat = ((o[keep] - e[keep]) ** 2 / e[keep]).sum()
    assert stat < chi2_crit(keep.sum() - 1), (stat, keep.sum())
    # members of the largest bucket: uniform, with replacement, the three positions independent
    b17 = np.asarray(buckets[17])
    in17 = tri[rot == 17]
    assert len(in17) > 5000
    pos = np.searchsorted(b17, in17)                             # index of every drawn match inside its bucket
    for col in range(3):
        c = np.bincount(pos[:, col], minlength=len(b17)).astype(np.float64)
        ee = len(in17) / len(b17)
        assert ((c - ee) ** 2 / ee).sum() < chi2_crit(len(b17) - 1)
    assert (pos[:, 0] == pos[:, 1]).mean() > 0.5 / len(b17)     # repeats do occur: with replacement (expected 1 / 200 of the draws)
    q = 10                                                       # coarse 10 x 10 table of (first, second)
    cell = (pos[:, 0] * q // len(b17)) * q + pos[:, 1] * q // len(b17)
    c = np.bincount(cell, minlength=q * q).astype(np.float64)
    ee = len(in17) / (q * q)
    assert ((c - ee) ** 2 / ee).sum() < chi2_crit(q * q - 1)
    # two-sample: the same number of draws made by numpy as the reference makes them
    np_state = np.random.get_state()
    np.random.seed(11)
    ref_rot = np.array([np.random.choice(range(60), p=prob) for _ in range(n_it)])
    np.random.set_state(np_state)
    a, b = np.bincount(rot, minlength=60).astype(np.float64), np.bincount(ref_rot, minlength=60).astype(np.float64)
    both = (a + b) >= 10
    stat2 = (((a - b) ** 2) / (a + b))[both].sum()
    assert stat2 < chi2_crit(both.sum() - 1), stat2
