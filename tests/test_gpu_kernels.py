"""GPU parity tests (-m gpu): every C-ABI entry point against the oracle and the golden vectors
captured from the reference.  Integer / index outputs must be bit-exact; fp32 network outputs
must agree to 1e-4 relative (the tolerance BASELINE.json's north_star states)."""
import os
import sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import yoho_oracle as orc  # noqa: E402
from yoho_amd import synth, weights as W  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4          # relative fp32 tolerance from BASELINE.json north_star


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) / max(np.max(np.abs(b)), 1e-30))


def rel_rows(a, b):
    """the worst ROW: every keypoint's (match's) error against that row's own magnitude - rel() above measures against the largest
    value of the whole array, which a row of small values can hide behind"""
    a, b = np.asarray(a, np.float64).reshape(len(a), -1), np.asarray(b, np.float64).reshape(len(b), -1)
    return float(np.max(np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), 1e-30)))


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ctx(hip, sd1, sd2):
    """context with the fp32-MFMA group conv (the bf16x3 default is exercised by ctx16 and the drop-in tests)"""
    c = hip.Context()
    c.load_partI(sd1)
    c.load_partII(sd2)
    c.set_gconv_mode("f32")
    c.set_partII_mode("f32")
    return c


def test_partI_golden(ctx, gold):
    g = gold("partI.npz")
    out = ctx.partI_forward(cu(g["x"]), want_inv=True, want_inv_np=True)
    eqv, inv = out["eqv"].cpu().numpy(), out["inv"].cpu().numpy()
    assert rel(eqv, g["eqv"]) < TOL and rel(inv, g["inv"]) < TOL
    assert rel_rows(eqv, g["eqv"]) < TOL and rel_rows(inv, g["inv"]) < TOL
    # numpy-order mean of OUR eqv must be bit-exact np.mean of our eqv
    assert np.array_equal(out["inv_np"].cpu().numpy(), np.mean(eqv, axis=-1))
    print("partI golden: rel err eqv %.3g inv %.3g" % (rel(eqv, g["eqv"]), rel(inv, g["inv"])))


@pytest.mark.parametrize("B", [1, 2, 31, 33, 100])
def test_partI_ragged_batches_vs_oracle(ctx, sd1, tables, B):
    x = synth.unit_features(B, seed=100 + B)
    out = ctx.partI_forward(cu(x))
    e, i = orc.partI_forward(x, sd1, tables.N)
    assert rel(out["eqv"].cpu().numpy(), e) < TOL and rel(out["inv"].cpu().numpy(), i) < TOL


def test_partI_equivariance_at_full_size(ctx, tables):
    # size-independent property at BASELINE's full size (5000 kp): PartI(x[:,:,P[i]]) == PartI(x)[:,:,P[i]].
    # Our kernel sums taps in a fixed slot order, so this holds to fp32 rounding, not bit-exactly.
    x = synth.unit_features(5000, seed=1)
    e0 = ctx.partI_forward(cu(x))["eqv"]
    P = torch.from_numpy(tables.P).cuda()
    for i in (17, 42):
        xi = cu(x)[:, :, P[i]].contiguous()
        ei = ctx.partI_forward(xi)["eqv"]
        err = (ei - e0[:, :, P[i]]).abs().max().item()
        assert err < 2e-5, err
    # unit norm over channels
    n = torch.linalg.norm(e0, dim=1)
    assert (n - 1).abs().max().item() < 1e-5


def _split_ctx(hip, sd1, sd2, mode):
    c = hip.Context()
    c.load_partI(sd1)
    c.load_partII(sd2)
    c.set_gconv_mode(mode)
    c.set_partII_mode(mode)
    return c


@pytest.fixture(scope="module")
def ctx16(hip, sd1, sd2):
    """second context running the PartI group conv on the bf16x3 split MFMA path"""
    return _split_ctx(hip, sd1, sd2, "bf16x3")


@pytest.fixture(scope="module")
def ctxh(hip, sd1, sd2):
    """third context: fp16x2 split MFMA path"""
    return _split_ctx(hip, sd1, sd2, "fp16x2")


@pytest.fixture
def ctx_of(ctx, ctx16, ctxh):
    return {"f32": ctx, "bf16x3": ctx16, "fp16x2": ctxh}


@pytest.mark.parametrize("mode,dmax", [("bf16x3", 5e-6), ("fp16x2", 2e-5)])
def test_partI_split_golden_and_vs_f32(ctx, ctx_of, mode, dmax, gold, sd1, tables):
    ctx16 = ctx_of[mode]
    g = gold("partI.npz")
    out = ctx16.partI_forward(cu(g["x"]), want_inv=True, want_inv_np=True)
    eqv, inv = out["eqv"].cpu().numpy(), out["inv"].cpu().numpy()
    print("%s golden: rel err eqv %.3g inv %.3g; worst row %.3g / %.3g" % (mode, rel(eqv, g["eqv"]), rel(inv, g["inv"]), rel_rows(eqv, g["eqv"]), rel_rows(inv, g["inv"])))
    assert rel(eqv, g["eqv"]) < TOL and rel(inv, g["inv"]) < TOL
    assert rel_rows(eqv, g["eqv"]) < TOL and rel_rows(inv, g["inv"]) < TOL
    assert np.array_equal(out["inv_np"].cpu().numpy(), np.mean(eqv, axis=-1))
    for B in (1, 2, 15, 16, 17, 33, 100):
        x = synth.unit_features(B, seed=200 + B)
        o = ctx16.partI_forward(cu(x))
        e, i = orc.partI_forward(x, sd1, tables.N)
        assert rel(o["eqv"].cpu().numpy(), e) < TOL and rel(o["inv"].cpu().numpy(), i) < TOL, B
    # input magnitudes far from 1 (tiny values exercise the low planes near the bottom of the 16-bit exponent range)
    for scale in (1e-3, 30.0):
        x = synth.unit_features(40, seed=77) * np.float32(scale)
        o = ctx16.partI_forward(cu(x))
        e, i = orc.partI_forward(x, sd1, tables.N)
        r = rel(o["eqv"].cpu().numpy(), e)
        print("%s input scale %g: rel err %.3g" % (mode, scale, r))
        assert r < TOL, scale
    # the two arithmetic paths agree to fp32 rounding at full size
    x = cu(synth.unit_features(5000, seed=1))
    e32 = ctx.partI_forward(x)["eqv"]
    e16 = ctx16.partI_forward(x)["eqv"]
    d = (e32 - e16).abs().max().item()
    print("%s vs f32 MFMA at 5000 kp: max abs diff %.3g" % (mode, d))
    assert d < dmax
    P = torch.from_numpy(tables.P).cuda()
    ei = ctx16.partI_forward(x[:, :, P[17]].contiguous())["eqv"]
    assert (ei - e16[:, :, P[17]]).abs().max().item() < 2e-5


@pytest.mark.parametrize("fmode", ["fourier", "fgemm", "fgemm256", "fgemm128"])
def test_partI_group_fourier_mode(hip, ctx, gold, sd1, tables, fmode):
    """modes 'fourier' / 'fgemm': the conv runs on group-Fourier coefficients (244 instead of 780 slab products),
    on fp32 MFMA, or with the two large layers as irrep GEMMs on the fp16x2 split MFMA"""
    c = hip.Context()
    c.load_partI(sd1)
    c.set_gconv_mode(fmode)
    g = gold("partI.npz")
    out = c.partI_forward(cu(g["x"]), want_inv=True, want_inv_np=True)
    eqv, inv = out["eqv"].cpu().numpy(), out["inv"].cpu().numpy()
    print("%s golden: rel err eqv %.3g inv %.3g; worst row %.3g / %.3g" % (fmode, rel(eqv, g["eqv"]), rel(inv, g["inv"]), rel_rows(eqv, g["eqv"]), rel_rows(inv, g["inv"])))
    assert rel(eqv, g["eqv"]) < TOL and rel(inv, g["inv"]) < TOL
    assert rel_rows(eqv, g["eqv"]) < TOL and rel_rows(inv, g["inv"]) < TOL
    for B in (1, 31, 33, 100, 257):
        x = synth.unit_features(B, seed=300 + B)
        o = c.partI_forward(cu(x))
        e, i = orc.partI_forward(x, sd1, tables.N)
        assert rel(o["eqv"].cpu().numpy(), e) < TOL and rel(o["inv"].cpu().numpy(), i) < TOL, B
    x = cu(synth.unit_features(5000, seed=1))
    d = (ctx.partI_forward(x)["eqv"] - c.partI_forward(x)["eqv"]).abs().max().item()
    print("%s vs direct f32 MFMA at 5000 kp: max abs diff %.3g" % (fmode, d))
    assert d < 1e-5


def test_partI_fp8_correction_mode(hip, ctx, gold, sd1, tables):
    """gconv mode 'fgemm8' (opt-in, round 5): the irrep GEMMs of the two large layers evaluate the main product a_h w_h on the fp16 pipe and
    BOTH correction products (a_h w_l + a_l w_h, 2^-11 of the main term) in fp8 e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4, two K16 steps per
    instruction - 2 / 3 of the matrix time.  The activation operand is scaled by the largest magnitude the transform in front of the GEMM
    wrote (a device word, no host round trip).  Tolerance: BASELINE's 1e-4; expected ~1.5e-5 (tools/fp8_correction_study.py: CPU emulation
    of exactly this arithmetic), asserted < 5e-5 so that a broken scale (corrections lost: 3e-4) cannot pass.  Deterministic; pair pass equal
    to the two single passes; and faster than the default mode on the 10 000-keypoint pair pass."""
    c = hip.Context()
    c.load_partI(sd1)
    c.set_gconv_mode("fgemm8")
    g = gold("partI.npz")
    out = c.partI_forward(cu(g["x"]), want_inv=True, want_inv_np=True)
    e_g, i_g = rel(out["eqv"].cpu().numpy(), g["eqv"]), rel(out["inv"].cpu().numpy(), g["inv"])
    print("fgemm8 golden: rel err eqv %.3g inv %.3g" % (e_g, i_g))
    assert e_g < 5e-5 and i_g < 5e-5
    worst = 0.0
    for B in (1, 31, 33, 100, 257, 1000):
        x = synth.unit_features(B, seed=300 + B)
        o = c.partI_forward(cu(x))
        e, i = orc.partI_forward(x, sd1, tables.N)
        worst = max(worst, rel(o["eqv"].cpu().numpy(), e), rel(o["inv"].cpu().numpy(), i))
    print("fgemm8 vs oracle, B = 1 .. 1000: worst rel err %.3g" % worst)
    assert worst < 5e-5
    x0, x1 = cu(synth.unit_features(5000, seed=41)), cu(synth.unit_features(4999, seed=42))
    d_ = c.partI_forward(x0)["eqv"]
    ref = ctx.partI_forward(x0)["eqv"]                                           # direct fp32 MFMA
    dd = (d_ - ref).abs().max().item()
    shipped = hip.Context()
    shipped.load_partI(sd1)
    ds = (shipped.partI_forward(x0)["eqv"] - ref).abs().max().item()
    print("at 5000 kp vs the direct fp32 kernel: fgemm8 max abs diff %.3g, default fgemm %.3g" % (dd, ds))
    assert dd < 5e-5 and ds < 1e-5 and dd > ds                                    # the new mode really runs a different arithmetic
    assert torch.equal(c.partI_forward(x0)["eqv"], d_)                            # deterministic
    # pair pass: one launch over both fragments; the activation scale is then the maximum over BOTH, so the bits may differ from the
    # single passes in the fp8 rounding of the corrections - the values may not
    pp = c.partI_forward_pair(x0, x1, want_inv=False, want_inv_np=True)
    s0, s1 = c.partI_forward(x0)["eqv"], c.partI_forward(x1)["eqv"]
    assert (pp["eqv"][:5000] - s0).abs().max().item() < 3e-5 and (pp["eqv"][5000:] - s1).abs().max().item() < 3e-5
    assert torch.equal(c.partI_forward_pair(x0, x1, want_inv=False, want_inv_np=True)["eqv"], pp["eqv"])
    # time of the GEMM launches of the pair pass, both modes (HIP events recorded by the library)
    ms = {}
    for name, cc in (("fgemm", shipped), ("fgemm8", c)):
        cc.set_profiling(True)
        for _ in range(4):
            cc.partI_forward_pair(x0, x1, want_inv=False, want_inv_np=True, check_range=False)
        torch.cuda.synchronize()
        ms[name] = [cc.kernel_ms(i) for i in range(4)] + [cc.kernel_ms(12)]
        cc.set_profiling(False)
    print("GEMM launches of a 9999-keypoint pair pass (ms): fgemm %s pass %.3f | fgemm8 %s pass %.3f" %
          ([round(v, 3) for v in ms["fgemm"][:4]], ms["fgemm"][4], [round(v, 3) for v in ms["fgemm8"][:4]], ms["fgemm8"][4]))
    assert ms["fgemm8"][1] < ms["fgemm"][1] and ms["fgemm8"][2] < ms["fgemm"][2]


def test_fgemm_tile_variants_are_bit_identical(hip, sd1):
    """the three GEMM blockings of the default mode (256 x 256 tile with eight waves, two per SIMD | 256 x 256 with four waves |
    256 x 128 tiles, two workgroups per CU) and the two transform kernels (two waves per SIMD | one) issue the same products in
    the same order per accumulator: identical bits, at ragged and full sizes and for a pair pass"""
    cs = []
    for m in ("fgemm", "fgemm256", "fgemm128"):
        c = hip.Context()
        c.load_partI(sd1)
        c.set_gconv_mode(m)
        cs.append(c)
    for B in (1, 33, 257, 1000, 5000):
        x = cu(synth.unit_features(B, seed=700 + B))
        outs = [c.partI_forward(x, want_inv=True, want_inv_np=True) for c in cs]
        for o in outs[1:]:
            assert all(torch.equal(outs[0][k], o[k]) for k in ("eqv", "inv", "inv_np")), B
    a, b = cu(synth.unit_features(5000, seed=41)), cu(synth.unit_features(4999, seed=42))
    ps = [c.partI_forward_pair(a, b, want_inv=False, want_inv_np=True) for c in cs]
    for p in ps[1:]:
        assert torch.equal(ps[0]["eqv"], p["eqv"]) and torch.equal(ps[0]["inv_np"], p["inv_np"])
    assert torch.isfinite(ps[0]["eqv"]).all()


def test_partI_depth_first_schedule_is_bit_identical(hip, sd1):
    """yoho_set_partI_schedule: the pass cut into chunks (one stream, or alternating over the caller's and the context's own
    stream) gives the bits of the breadth-first pass - ragged sizes, chunks that straddle the fragment boundary of a pair pass,
    work queued on the caller's stream before and after the call - and the profiling hook sums over the chunks"""
    c = hip.Context()
    c.load_partI(sd1)
    x = cu(synth.unit_features(1500, seed=51))
    a, b = cu(synth.unit_features(1100, seed=52)), cu(synth.unit_features(700, seed=53))
    ref = c.partI_forward(x, want_inv=True, want_inv_np=True)
    refp = c.partI_forward_pair(a, b, want_inv=True, want_inv_np=True)
    for chunk, ns in ((256, 1), (256, 2), (512, 2), (1000, 1), (1024, 2), (4096, 2)):
        c.set_partI_schedule(chunk, ns)
        o = c.partI_forward(x.clone(), want_inv=True, want_inv_np=True)
        p = c.partI_forward_pair(a, b, want_inv=True, want_inv_np=True)
        s = p["eqv"].sum()                                   # consumer on the caller's stream right behind the call
        assert all(torch.equal(ref[k], o[k]) for k in ("eqv", "inv", "inv_np")), (chunk, ns)
        assert all(torch.equal(refp[k], p[k]) for k in ("eqv", "inv", "inv_np")), (chunk, ns)
        assert torch.isfinite(s)
    c.set_partI_schedule(512, 2)
    c.set_profiling(True)
    c.partI_forward(x, want_inv=False, want_inv_np=True)
    ms = [c.kernel_ms(i) for i in range(13)]
    c.set_profiling(False)
    assert all(m > 0 for m in ms[:4]) and ms[12] > 0 and ms[6] > 0
    c.set_partI_schedule(0, 1)
    with pytest.raises(RuntimeError):
        c.set_partI_schedule(1024, 3)
    # the clock probe (bench.py's power section): a plausible shader clock
    t = c.clock_probe(2000)
    torch.cuda.synchronize()
    cyc, ticks, khz = (int(v) for v in t.cpu())
    mhz = cyc / ticks * khz / 1000.0
    print("clock probe: %.0f MHz (%d cycles / %d ticks of a %d kHz counter)" % (mhz, cyc, ticks, khz))
    assert ticks > 0 and 100.0 < mhz < 3000.0


def test_lane_streams_are_picked_by_measured_overlap(hip):
    """hip.streams_overlap / concurrent_stream (the backbone lanes): a stream trivially does not overlap itself, a picked stream
    overlaps every stream it was picked against, and is never the null stream"""
    H = hip
    c = H.get_context()
    cur = torch.cuda.current_stream()
    assert not H.streams_overlap(c, cur, cur)
    a = H.concurrent_stream(c, [cur])
    b = H.concurrent_stream(c, [cur, a])
    assert a.cuda_stream != 0 and b.cuda_stream != 0 and a.cuda_stream != b.cuda_stream
    assert H.streams_overlap(c, cur, a) and H.streams_overlap(c, cur, b) and H.streams_overlap(c, a, b)


def test_contexts_with_different_group_tables_coexist(hip, sd1, tables, tmp_path):
    """The slot tables of the direct-conv kernels travel in the launch arguments (device memory of the context), so a second
    context on a RELABELLED copy of the group (element a <-> pi[a], identity fixed) lives beside the default one: its PartI on the
    relabelled input is the default context's output relabelled, in the direct fp32 mode and in the default Fourier mode, and the
    default context still answers as before (round 2 refused the second table set)."""
    rs = np.random.RandomState(5)
    pi = np.concatenate([[0], 1 + rs.permutation(59)])
    inv = np.argsort(pi)
    d = tmp_path / "so3"
    os.makedirs(d)
    np.save(d / "Rotation.npy", tables.R64[pi])
    np.save(d / "Nei_Index_in_SO3_ordered_13.npy", inv[tables.N[pi]].astype(np.float64))
    np.save(d / "60_60.npy", inv[tables.P[pi][:, pi]].astype(np.float64))
    x = synth.unit_features(70, seed=91)
    for mode, tol in (("f32", 2e-5), ("fgemm", 2e-5)):
        a = hip.Context()
        a.load_partI(sd1); a.set_gconv_mode(mode)
        ref = a.partI_forward(cu(x), want_inv=True)
        b = hip.Context(so3_dir=str(d))
        b.load_partI(sd1); b.set_gconv_mode(mode)
        ob = b.partI_forward(cu(np.ascontiguousarray(x[:, :, pi])), want_inv=True)
        again = a.partI_forward(cu(x), want_inv=True)
        assert torch.equal(again["eqv"], ref["eqv"]), mode                      # the first context is undisturbed
        e = ref["eqv"].cpu().numpy()
        assert np.abs(ob["eqv"].cpu().numpy() - e[:, :, pi]).max() < tol, mode  # relabelled group, relabelled answer
        assert np.abs(ob["inv"].cpu().numpy() - ref["inv"].cpu().numpy()).max() < tol, mode


def test_group_mean_np_bitexact(ctx):
    x = synth.unit_features(333, seed=5) * np.float32(1.7)
    out = ctx.group_mean_np(cu(x)).cpu().numpy()
    assert np.array_equal(out, np.mean(x, axis=-1))


def test_nn_search_bitexact(ctx, gold):
    g = gold("pdist.npz")
    d, i = ctx.nn_search(cu(g["A"]), cu(g["B"]))
    ref = orc.pdist_l2(g["A"], g["B"])
    assert np.array_equal(i.cpu().numpy(), np.argmin(ref, 1))
    assert np.array_equal(d.cpu().numpy(), ref.min(1))
    # reference's own distances (torch-CPU sqrt is 1 ulp low in <2% of cases): same argmin
    assert np.array_equal(i.cpu().numpy(), np.argmin(g["dist"], 1))
    # 3-D variant (simple_yoho/yoho_extract.py:33-39 feature transfer)
    rs = np.random.RandomState(0)
    q, s = rs.rand(777, 3).astype(np.float32), rs.rand(4001, 3).astype(np.float32)
    d3, i3 = ctx.nn_search(cu(q), cu(s))
    assert np.array_equal(i3.cpu().numpy(), np.argmin(orc.pdist_l2(q, s), 1))


def test_nn_ties_first_index(ctx):
    a = np.zeros((5, 32), np.float32)
    b = np.ones((700, 32), np.float32)
    b[[3, 300, 699]] = 0.5
    _, i = ctx.nn_search(cu(a), cu(b))
    assert (i.cpu().numpy() == 3).all()


def test_mutual_match_golden_and_full_size(ctx, gold):
    g = gold("chain.npz")
    m = ctx.mutual_nn(cu(g["inv0"]), cu(g["inv1"]))
    assert m.dtype == torch.int64 and np.array_equal(m.cpu().numpy(), g["match"])
    # full size (5000 x 5000): compare with the oracle, plus structural properties
    pr = synth.make_pair(5000, seed=2)
    i0, i1 = np.mean(pr["feat0"], -1), np.mean(pr["feat1"], -1)
    mm = ctx.mutual_nn(cu(i0), cu(i1)).cpu().numpy()
    assert np.array_equal(mm, orc.mutual_match(i0, i1))
    assert (np.diff(mm[:, 0]) > 0).all() and len(set(mm[:, 1])) == len(mm)


def test_mutual_nn_prefilter_equals_brute_force(hip):
    """yoho_mutual_nn through the fp16-MFMA pre-filter + exact candidates (matchf.hip) against the brute-force kernels: the SAME
    match list on descriptor-like data, exact duplicates and near-duplicates (ties go to the lowest index), tiny / large / mixed
    magnitudes, zero rows, ragged sizes, and inputs the pre-filter has to decline (NaN, inf, beyond the fp16 range)"""
    c = hip.Context()
    rs = np.random.RandomState(0)

    def both(a, b):
        ad, bd = cu(a), cu(b)
        c.set_nn_prefilter(True)
        m1 = c.mutual_nn(ad, bd).cpu().numpy()
        c.set_nn_prefilter(False)
        m0 = c.mutual_nn(ad, bd).cpu().numpy()
        c.set_nn_prefilter(True)
        return m1, m0

    cases = {}
    pr = synth.make_pair(5000, seed=2)
    cases["descriptor-like 5000 x 5000"] = (np.mean(pr["feat0"], -1), np.mean(pr["feat1"], -1))
    a = rs.randn(1500, 32).astype(np.float32) * 0.2
    b = np.concatenate([a[rs.permutation(1500)[:1000]], a[:700] + rs.randn(700, 32).astype(np.float32) * 1e-6,
                        a[300:900] + rs.randn(600, 32).astype(np.float32) * 1e-4, a[:701] + rs.randn(701, 32).astype(np.float32) * 1e-3])
    cases["duplicates and near-duplicates 1500 x 3001"] = (a, np.ascontiguousarray(b))
    cases["tiny magnitudes"] = (a * np.float32(1e-4), np.ascontiguousarray(b) * np.float32(1e-4))
    cases["large magnitudes"] = (a * np.float32(50), np.ascontiguousarray(b) * np.float32(50))
    sc = np.exp(rs.randn(1500, 1) * 2).astype(np.float32)
    cases["mixed norms"] = (a * sc, np.ascontiguousarray(b) * np.exp(rs.randn(3001, 1) * 2).astype(np.float32))
    z = rs.randn(1200, 32).astype(np.float32)
    z[::7] = 0
    cases["zero rows"] = (z, np.ascontiguousarray(z[::-1]))
    # every pair a candidate: the candidate list overflows and the brute-force kernels take over (ties -> lowest index)
    one = rs.randn(1, 32).astype(np.float32)
    cases["all rows identical 1100 x 1300 (candidate list full)"] = (np.repeat(one, 1100, 0), np.repeat(one, 1300, 0))
    few = rs.randn(5, 32).astype(np.float32)
    cases["five distinct rows repeated 1500 x 1500"] = (few[rs.randint(0, 5, 1500)], few[rs.randint(0, 5, 1500)])
    for name, (x, y) in cases.items():
        m1, m0 = both(x, y)
        assert m1.shape == m0.shape and np.array_equal(m1, m0), name
        print("mutual NN pre-filter, %s: %d matches, identical to brute force" % (name, len(m1)))
    big = rs.randn(1100, 32).astype(np.float32)
    for bad in (np.nan, np.inf, 7e4):
        x = big.copy()
        x[17, 3] = bad
        m1, m0 = both(x, big[::-1].copy())
        assert np.array_equal(m1, m0), bad
    # and against the oracle on one case
    x, y = cases["duplicates and near-duplicates 1500 x 3001"]
    assert np.array_equal(both(x, y)[0], orc.mutual_match(x, y))


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "fp16x2"])
def test_des2r_golden(ctx_of, mode, gold, sd1, tables):
    ctx = ctx_of[mode]
    g = gold("chain.npz")
    pr = synth.make_pair(int(g["K"]), seed=int(g["pair_seed"]))
    e0 = ctx.partI_forward(cu(pr["feat0"]))["eqv"]
    e1 = ctx.partI_forward(cu(pr["feat1"]))["eqv"]
    m = torch.from_numpy(g["match"]).cuda()
    idx, cor = ctx.des2r(e1[m[:, 1]].contiguous(), e0[m[:, 0]].contiguous(), want_cor=True)
    assert rel(cor.cpu().numpy(), g["cor"]) < TOL
    top2 = np.sort(g["cor"], 1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 1e-4                 # away from near-ties the index must be identical
    assert np.array_equal(idx.cpu().numpy()[safe], g["dr_index"][safe]) and safe.mean() > 0.9
    # recovers the planted rotation (SURVEY section 4 property 2) on exactly permuted inputs
    x = synth.unit_features(64, seed=3)
    ea = ctx.partI_forward(cu(x))["eqv"]
    for i in (0, 13, 59):
        eb = ctx.partI_forward(cu(np.ascontiguousarray(x[:, :, tables.P[i]])))["eqv"]
        assert (ctx.des2r(ea, eb).cpu().numpy() == i).all()


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "fp16x2"])
def test_partII_hyp_golden(ctx_of, mode, gold, sd1, sd2, tables):
    ctx = ctx_of[mode]
    g = gold("chain.npz")
    pr = synth.make_pair(int(g["K"]), seed=int(g["pair_seed"]))
    m, dr = g["match"], g["dr_index"]
    e0 = orc.partI_extract(pr["feat0"], sd1, tables.N, batch=40)
    e1 = orc.partI_extract(pr["feat1"], sd1, tables.N, batch=40)
    b = orc.batch_create(pr["feat0"][m[:, 0]], pr["feat1"][m[:, 1]], e0[m[:, 0]], e1[m[:, 1]], dr)
    args = [cu(b[k]) for k in ("before_eqv0", "before_eqv1", "after_eqv0", "after_eqv1")]
    keep = [a.clone() for a in args]
    q = ctx.partII_forward(*args, cu(dr))
    assert all(torch.equal(a, k) for a, k in zip(args, keep)), "inputs must not be modified"
    qo = orc.partII_forward(b["before_eqv0"], b["before_eqv1"], b["after_eqv0"], b["after_eqv1"], dr, sd2, tables.N, tables.P)
    print("partII %s: rel err vs oracle %.3g (M=%d)" % (mode, rel(q.cpu().numpy(), qo), len(dr)))
    assert rel(q.cpu().numpy(), qo) < TOL
    assert rel(q.cpu().numpy()[:16], g["quat16"]) < TOL
    k0, k1 = pr["keys0"][m[:, 0]], pr["keys1"][m[:, 1]]
    # hypotheses from the REFERENCE's quaternions must reproduce the reference's Trans_pre to f64 rounding
    T = ctx.hyp_from_quat(cu(qo), cu(dr), cu(k0), cu(k1)).cpu().numpy()
    To = orc.hyp_from_quat(qo, dr, k0, k1, tables.R32)
    assert np.allclose(T, To, rtol=0, atol=1e-12)
    T2 = ctx.hyp_from_quat(q, cu(dr), cu(k0), cu(k1)).cpu().numpy()
    assert rel(T2, g["trans_pre"]) < TOL


@pytest.mark.parametrize("M", [1, 17, 40, 100])
def test_partII_ragged_match_counts(ctx, ctxh, sd2, tables, M):
    """odd tile counts (16- and 32-match tiles, 256-column GEMM padding): default fp16x2 / Fourier first layer vs oracle and fp32"""
    rs = np.random.RandomState(500 + M)
    mk = lambda sd: synth.unit_features(M, seed=sd)
    a, b, c_, d = mk(1), mk(2), mk(3), mk(4)
    dr = rs.randint(0, 60, size=M).astype(np.int64)
    args = [cu(x) for x in (a, b, c_, d)]
    q = ctxh.partII_forward(*args, cu(dr)).cpu().numpy()
    q32 = ctx.partII_forward(*args, cu(dr)).cpu().numpy()
    qo = orc.partII_forward(a, b, c_, d, dr, sd2, tables.N, tables.P)
    assert q.shape == (M, 4) and np.isfinite(q).all()
    assert rel(q, qo) < TOL and rel(q32, qo) < TOL, (rel(q, qo), rel(q32, qo))


@pytest.mark.parametrize("mode,tol", [("cgemm", TOL), ("cgemm8", TOL)])
def test_partII_cone_gemm_modes(hip, ctx, ctxh, sd1, sd2, tables, mode, tol):
    """PartII modes 3 / 4: the 13-element cone layer as one implicit GEMM (cgemm_kernel: B-operand stage blocks picked per tap from the 45
    cone elements the inverse transform leaves, fp16x2 products or fp16 main product + fp8 correction products) against the oracle, the
    fp32 kernels and the direct cone kernel of mode 2 - at match counts around every tile boundary of the new kernels (32-match transform
    tiles, 16-match cone1 tiles, 256-match column tiles: the last column tile of a pass is ragged and its unwritten columns must not
    leak), through the plain and the row-indexed entry, and repeated (workspace reuse with stale column tiles from a larger pass)."""
    c = hip.Context()
    c.load_partI(sd1)
    c.load_partII(sd2)
    c.set_partII_mode(mode)
    assert c.supports_matched()
    worst = 0.0
    feats = [synth.unit_features(700, seed=sd) for sd in (11, 12, 13, 14)]
    for M in (700, 1, 17, 40, 255, 256, 257, 300):           # the large pass first: the small ones then see its stale stage blocks
        rs = np.random.RandomState(900 + M)
        a, b, c_, d = (np.ascontiguousarray(x[700 - M:]) for x in feats)
        dr = rs.randint(0, 60, size=M).astype(np.int64)
        args = [cu(x) for x in (a, b, c_, d)]
        q = c.partII_forward(*args, cu(dr)).cpu().numpy()
        qd = ctxh.partII_forward(*args, cu(dr)).cpu().numpy()     # the direct cone kernel (pinned to the oracle by the tests above)
        assert q.shape == (M, 4) and np.isfinite(q).all()
        assert rel(q, qd) < tol, (M, rel(q, qd))
        if M in (1, 40, 257):                                     # and the oracle itself around the tile boundaries
            qo = orc.partII_forward(a, b, c_, d, dr, sd2, tables.N, tables.P)
            worst = max(worst, rel(q, qo))
            assert rel(q, qo) < tol, (M, rel(q, qo))
        assert np.array_equal(q, c.partII_forward(*args, cu(dr)).cpu().numpy()), M          # same bits on a repeat
    print("partII %s: worst rel err vs oracle %.3g" % (mode, worst))
    # the row-indexed entry (what run_pair / yoho_register_pair use): rows addressed through a match list
    K = 300
    pr = synth.make_pair(K, seed=21)
    f0, f1 = cu(pr["feat0"]), cu(pr["feat1"])
    o0, o1 = c.partI_forward(f0)["eqv"], c.partI_forward(f1)["eqv"]
    rs = np.random.RandomState(3)
    match = torch.from_numpy(np.stack([rs.permutation(K)[:180], rs.permutation(K)[:180]], 1).astype(np.int64)).cuda()
    dr = torch.from_numpy(rs.randint(0, 60, size=180).astype(np.int64)).cuda()
    qi = c.partII_forward_matched(f0, f1, o0, o1, match, dr)
    m0, m1 = match[:, 0], match[:, 1]
    qg = c.partII_forward(f1[m1].contiguous(), f0[m0].contiguous(), o1[m1].contiguous(), o0[m0].contiguous(), dr)
    assert torch.equal(qi, qg)
    assert c.range_status() == (False, False)


def test_partII_zero_matches(ctxh):
    e = torch.empty((0, 32, 60), dtype=torch.float32, device="cuda")
    q = ctxh.partII_forward(e, e, e, e, torch.empty((0,), dtype=torch.int64, device="cuda"))
    assert tuple(q.shape) == (0, 4)


def test_partI_large_batch_is_chunked_consistently(hip, sd1):
    """more keypoints than one pass takes (16384): same rows as separate calls, in the default arithmetic mode"""
    c = hip.Context()
    c.load_partI(sd1)
    B = 16384 + 300
    x = cu(synth.unit_features(B, seed=9))
    full = c.partI_forward(x, want_inv=True, want_inv_np=True)
    for lo, hi in ((0, 64), (16384 - 32, 16384 + 64), (B - 50, B)):
        part = c.partI_forward(x[lo:hi].contiguous(), want_inv=True, want_inv_np=True)
        for k in ("eqv", "inv", "inv_np"):
            d = (full[k][lo:hi] - part[k]).abs().max().item()
            assert d < 2e-6, (k, lo, d)
    n = torch.linalg.norm(full["eqv"], dim=1)
    assert (n - 1).abs().max().item() < 1e-5


def test_matched_variants_equal_gathered(ctxh, tables):
    """row-indexed Des2R / PartII (HBM-resident pipeline) == the same calls on gathered rows"""
    K, M = 300, 130
    rs = np.random.RandomState(11)
    f0, f1 = cu(synth.unit_features(K, seed=21)), cu(synth.unit_features(K, seed=22))
    e0, e1 = cu(synth.unit_features(K, seed=23)), cu(synth.unit_features(K, seed=24))
    match = torch.from_numpy(np.stack([rs.permutation(K)[:M], rs.permutation(K)[:M]], 1).astype(np.int64)).cuda()
    m0, m1 = match[:, 0], match[:, 1]
    dr_g = ctxh.des2r(e1[m1].contiguous(), e0[m0].contiguous())
    dr_m = ctxh.des2r_matched(e1, e0, match)
    assert torch.equal(dr_g, dr_m)
    q_g = ctxh.partII_forward(f1[m1].contiguous(), f0[m0].contiguous(), e1[m1].contiguous(), e0[m0].contiguous(), dr_g)
    q_m = ctxh.partII_forward_matched(f0, f1, e0, e1, match, dr_g)
    assert torch.equal(q_g, q_m)


def test_partI_forward_pair_equals_concatenated(hip, ctxh, sd1):
    c = hip.Context()                                       # default arithmetic mode
    c.load_partI(sd1)
    a, b = cu(synth.unit_features(70, seed=41)), cu(synth.unit_features(45, seed=42))
    o1 = c.partI_forward(torch.cat([a, b]), want_inv=True, want_inv_np=True)
    o2 = c.partI_forward_pair(a, b, want_inv=True, want_inv_np=True)
    assert all(torch.equal(o1[k], o2[k]) for k in ("eqv", "inv", "inv_np"))
    with pytest.raises(RuntimeError):
        ctxh.partI_forward_pair(a, b)                       # direct-conv modes: the caller concatenates (pipeline.run_pair does)


def test_quat2mat_bitexact(ctx, gold):
    g = gold("quat.npz")
    M = g["q"].shape[0]
    z = np.zeros((M, 3))
    idx = np.zeros(M, np.int64)                      # group element 0 = identity
    T = ctx.hyp_from_quat(cu(g["q"]), cu(idx), cu(z), cu(z)).cpu().numpy()
    R0 = np.load(os.path.join(ctx.tables.dir, "Rotation.npy")).astype(np.float32)[0].astype(np.float64)
    assert np.allclose(T[:, :, :3], g["mats"] @ R0, rtol=0, atol=1e-15)


def test_yohoo_golden_and_micro(ctx, gold):
    g = gold("chain.npz")
    pr = synth.make_pair(int(g["K"]), seed=int(g["pair_seed"]))
    m = g["match"]
    k0, k1, T = pr["keys0"][m[:, 0]], pr["keys1"][m[:, 1]], g["trans_pre"]
    for seed, it, rec, tr in ((1234, 1000, "yohoo_recall", "yohoo_trans"), (4321, 20, "yohoo20_recall", "yohoo20_trans")):
        np.random.seed(seed)
        order = np.arange(T.shape[0]); np.random.shuffle(order)
        H = min(it, T.shape[0])
        res, counts = ctx.o_score(cu(k0), cu(k1), cu(T), cu(order), H, 0.09)
        bh, bc = res.cpu().numpy()
        assert bh == int(g[rec]) and np.array_equal(T[order[bh]], g[tr])
        ref_counts = [orc.inlier_count(k0, k1, T[order[h]], 0.09) for h in range(H)]
        assert np.array_equal(counts.cpu().numpy(), ref_counts)
    # estimator micro-benchmark size (M=1500, H=1000)
    ec = synth.estimator_case(1500, 1000, seed=4)
    order = np.arange(1500)[::-1].copy()
    res, counts = ctx.o_score(cu(ec["k0"]), cu(ec["k1"]), cu(ec["T"]), cu(order), 1000, 0.09)
    bid, cnt, Tb = orc.yohoo_select(ec["k0"], ec["k1"], ec["T"], order, 0.09, 1000)
    assert tuple(res.cpu().numpy()) == (bid, cnt)
    # nothing beats zero -> count 0 (caller keeps eye(4))
    far = ec["T"].copy(); far[:, :, 3] += 1e3
    res, _ = ctx.o_score(cu(ec["k0"]), cu(ec["k1"]), cu(far), None, 50, 0.09)
    assert res.cpu().numpy()[1] == 0


def test_kabsch_and_yohoc(ctx, gold):
    g = gold("kabsch.npz")
    n = g["k0"].shape[0]
    k0 = g["k0"].reshape(n * 3, 3); k1 = g["k1"].reshape(n * 3, 3)
    tri = np.arange(n * 3, dtype=np.int64).reshape(n, 3)
    dets = np.array([np.linalg.det(T[:, :3]) for T in g["T"]])
    refl = (dets < 0).astype(np.uint8)
    assert 0 < refl.sum() < n
    _, _, T_all, _ = ctx.c_ransac(cu(k0), cu(k1), cu(tri), cu(refl), 0.07, want_all=True)
    assert np.allclose(T_all.cpu().numpy(), g["T"], rtol=0, atol=1e-9)          # incl. the reference's reflections
    _, _, T_prop, _ = ctx.c_ransac(cu(k0), cu(k1), cu(tri), None, 0.07, want_all=True)
    Tp = T_prop.cpu().numpy()
    assert np.allclose(np.linalg.det(Tp[:, :, :3]), 1, atol=1e-9)
    for i in range(n):
        To, _ = orc.threepps2tran(g["k0"][i], g["k1"][i], proper=True)
        assert np.allclose(Tp[i], To, rtol=0, atol=1e-9)
    # full YOHO-C on the golden chain with the reference's RNG stream
    c = gold("chain.npz")
    pr = synth.make_pair(int(c["K"]), seed=int(c["pair_seed"]))
    m, dr = c["match"], c["dr_index"]
    kk0, kk1 = pr["keys0"][m[:, 0]], pr["keys1"][m[:, 1]]
    np.random.seed(99)
    tri = orc.yohoc_draw_triples(dr, 200, np.random)
    it, cnt, Tc, dets = orc.yohoc_select(kk0, kk1, tri, 0.07)
    best_T, res, T_all, counts = ctx.c_ransac(cu(kk0), cu(kk1), cu(tri), cu((dets < 0).astype(np.uint8)), 0.07, want_all=True)
    nodup = np.array([len(set(t)) == 3 for t in tri])
    ref_counts = np.array([orc.inlier_count(kk0, kk1, orc.threepps2tran(kk0[t], kk1[t])[0], 0.07) for t in tri])
    assert np.array_equal(counts.cpu().numpy()[nodup], ref_counts[nodup])
    bi, bc = res.cpu().numpy()
    assert bi == it == int(c["yohoc_recall"]) and bc == cnt
    assert np.allclose(best_T.cpu().numpy(), c["yohoc_trans"], rtol=0, atol=1e-9)


def test_group_gather(ctx, tables):
    rs = np.random.RandomState(1)
    K, n = 500, 6000
    keys = rs.rand(K, 3) * 2
    out = torch.zeros((K, 32, 60), dtype=torch.float32, device="cuda")
    pts = [(rs.rand(n + g, 3) * 2).astype(np.float32) for g in range(60)]
    feats = [rs.randn(n + g, 32).astype(np.float32) for g in range(60)]
    for g in range(60):
        idx = ctx.group_gather(cu(keys), cu(pts[g]), cu(feats[g]), g, out, want_idx=(g == 7))
        if g == 7:
            assert np.array_equal(idx.cpu().numpy(), orc.group_gather_one(keys, pts[g], feats[g], tables.R64[g])[1])
    assert np.array_equal(out.cpu().numpy(), orc.group_gather(keys, pts, feats, tables.R64))


def test_group_transfer_batch_equals_per_copy_calls(hip, tables):
    """yoho_group_transfer_batch (the feature-transfer body of the 60-rotation loop for the copies of one backbone pass) = the three
    calls per copy it replaces, bit for bit, with and without the hash grid; ragged copy sizes"""
    c = hip.Context()
    rs = np.random.RandomState(3)
    pts = cu(rs.rand(4000, 3) * 1.5)
    kidx = cu(rs.permutation(4000)[:300].astype(np.int64))
    Rs = [tables.R64[g] for g in (0, 7, 33, 59)]
    ds = [cu((rs.rand(m, 3) * 1.5).astype(np.float32)) for m in (3500, 1200, 900, 2048)]
    ft = [cu(rs.randn(d.shape[0], 32).astype(np.float32)) for d in ds]
    for cell in (0.0, 0.05):
        c.set_nn_grid(cell)
        ref = torch.zeros((300, 32, 60), dtype=torch.float32, device="cuda")
        for j in range(4):
            q = c.rotate_select(pts, Rs[j], kidx)
            _, idx = c.nn_search(q, ds[j], want_dist=False, squared=True)
            c.group_scatter(ft[j], idx, 10 + j, ref)
        out = torch.zeros_like(ref)
        c.group_transfer_batch(pts, kidx, Rs, ds, ft, 10, out)
        assert torch.equal(out, ref), cell
    c.set_nn_grid(0)
    with pytest.raises(ValueError):
        c.group_transfer_batch(pts, kidx, Rs, ds[:3], ft, 10, out)
    with pytest.raises(RuntimeError):
        c.group_transfer_batch(pts, kidx, Rs, ds, ft, 58, out)          # 58 + 4 copies > 60 group elements


def test_grid_nn_equals_brute_force(hip, tables):
    """yoho_set_nn_grid changes the search, never the answer: both fp32 distance types and the f64 group gather, for a
    well-chosen cell, one that is far too small (most queries fall back to brute force) and a coarse one (long cell lists);
    with far-away queries, duplicated targets (lowest index wins) and queries sitting exactly on targets."""
    c = hip.Context()
    rs = np.random.RandomState(3)
    cloud = synth.surface_cloud(60000, seed=4, extent=2.0)
    vox = np.floor(cloud / 0.025).astype(np.int64)
    _, first = np.unique(vox, axis=0, return_index=True)
    tgt = cloud[np.sort(first)].astype(np.float32)                       # one point per voxel, as the backbone's down-sampling leaves it
    tgt = np.concatenate([tgt, tgt[:300]])                               # duplicates: ties on distance
    src = np.concatenate([cloud[rs.permutation(len(cloud))[:3000]], tgt[100:400].astype(np.float64),
                          rs.rand(200, 3) * 2 + 5.0, rs.rand(200, 3) * 2.0]).astype(np.float32)
    s_d, t_d = cu(src), cu(tgt)
    assert src.shape[0] * tgt.shape[0] >= 1 << 20
    ref = {}
    for squared in (False, True):
        d, i = c.nn_search(s_d, t_d, squared=squared)
        ref[squared] = (d.cpu().numpy(), i.cpu().numpy())
    keys = np.concatenate([cloud[rs.permutation(len(cloud))[:2000]], rs.rand(100, 3) * 9.0])
    feat = rs.randn(len(tgt), 32).astype(np.float32)
    out0 = torch.zeros((len(keys), 32, 60), dtype=torch.float32, device="cuda")
    gi0 = {g: c.group_gather(cu(keys), t_d, cu(feat), g, out0, want_idx=True).cpu().numpy() for g in (0, 7, 33)}
    try:
        for cell in (0.025, 0.004, 0.2):
            c.set_nn_grid(cell)
            for squared in (False, True):
                d, i = c.nn_search(s_d, t_d, squared=squared)
                assert np.array_equal(i.cpu().numpy(), ref[squared][1]), (cell, squared)
                assert np.array_equal(d.cpu().numpy(), ref[squared][0]), (cell, squared)
            out1 = torch.zeros_like(out0)
            for g in (0, 7, 33):
                gi = c.group_gather(cu(keys), t_d, cu(feat), g, out1, want_idx=True).cpu().numpy()
                assert np.array_equal(gi, gi0[g]), (cell, g)
            assert torch.equal(out1, out0)
        with pytest.raises(hip.YohoError):
            c.set_nn_grid(-1.0)
    finally:
        c.set_nn_grid(0)


def test_error_paths(hip):
    c = hip.Context()
    x = torch.zeros((4, 32, 60), device="cuda")
    with pytest.raises(hip.YohoError, match="not loaded"):
        c.partI_forward(x)
    with pytest.raises(ValueError):
        c.partI_forward(torch.zeros((4, 32, 59), device="cuda"))
    with pytest.raises(TypeError):
        c.group_mean_np(torch.zeros((4, 32, 60)))
