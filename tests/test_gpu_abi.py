"""The C ABI driven directly (-m gpu): every entry point of include/yoho_hip.h with arguments it must refuse.

The Python wrapper (yoho_amd/hip.py) validates shapes before it calls the library, so none of the other tests ever shows the library
a bad argument.  A binding written against the header (the cgo / JNI / cffi stub of INTEGRATION.md) has no such wrapper: these tests
call the exported symbols through ctypes with a NULL context, NULL required pointers, negative counts, counts beyond a stated
capacity, pointers that are not aligned as the header demands, a wrong arithmetic mode for the *_pair / *_indexed entries, and
weights that were never loaded - and check the YOHO_E* code, that `yoho_last_error()` names the entry point, and that the context
still works afterwards.  Empty inputs (M = 0, K = 0, n = 0) are valid wherever the header says so and must launch nothing.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from yoho_amd import synth, weights as W

pytestmark = pytest.mark.gpu
EINVAL, ENOWEIGHTS = -1, -3


@pytest.fixture(scope="module")
def env(hip, sd1, sd2):
    lib = hip.load_library()
    ctx = hip.Context()
    ctx.load_partI(sd1)
    ctx.load_partII(sd2)
    ctx.load_fcgf(W.synth_state_dict(W.FCGF_SPEC, 3))
    bare = hip.Context()                                       # no weights at all
    return lib, ctx, bare


def err(lib):
    return lib.yoho_last_error().decode()


def test_every_entry_point_refuses_bad_arguments(env, hip):
    lib, ctx, bare = env
    h, hb = ctx._h, bare._h
    dev = "cuda"
    f32 = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    f64 = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)
    i64 = lambda *s: torch.zeros(s, dtype=torch.int64, device=dev)
    i32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
    x = torch.from_numpy(synth.unit_features(8, seed=1)).cuda()
    eqv, inv, invnp = f32(8, 32, 60), f32(8, 32), f32(8, 32)
    k0, k1, T = f64(8, 3), f64(8, 3), f64(8, 3, 4)
    idx, quat, cnt = i64(8), f32(8, 4), i32(4)
    pairs = i64(8, 2)
    p = lambda t: C.c_void_p(t.data_ptr())
    off = lambda t, b: C.c_void_p(t.data_ptr() + b)            # a pointer `b` bytes into a tensor: misaligned for b = 4
    N = None                                                   # NULL
    one = C.c_int(0)
    cfg = hip.FcgfConfig()
    res = hip.PairResultC()
    hostR = (C.c_double * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    offs2 = (C.c_int32 * 2)(0, 4)
    offs_bad = (C.c_int32 * 3)(0, 4, 2)
    ptr_arr = (C.c_void_p * 1)(x.data_ptr())
    m_arr = (C.c_int * 1)(8)
    ms = C.c_float(0)

    # (entry point, arguments, expected code, text the message must contain)
    # host-side arguments of yoho_c_draw_np (numpy arrays kept alive by these names)
    _key = np.zeros(624, np.uint32); _prob = np.full(60, 1.0 / 60); _start = np.arange(61, dtype=np.int64) * 2
    _start_bad = _start.copy(); _start_bad[7] = 99
    _members = np.arange(120, dtype=np.int64); _tri = np.zeros((4, 3), np.int64)
    hkey, hprob, hstart, hstart_bad, hmembers, htri = (C.c_void_p(a_.ctypes.data) for a_ in (_key, _prob, _start, _start_bad, _members, _tri))
    mtpos, badpos = C.c_int(624), C.c_int(700)
    cases = [
        ("yoho_ctx_create", (0, N, N, N, C.byref(C.c_void_p())), EINVAL, "yoho_ctx_create"),
        ("yoho_load_partI", (N, N), EINVAL, "yoho_load_partI"),
        ("yoho_load_partI", (h, N), EINVAL, "yoho_load_partI"),
        ("yoho_load_partII", (h, N), EINVAL, "yoho_load_partII"),
        ("yoho_partI_forward", (N, p(x), 8, p(eqv), N, N, N), EINVAL, "yoho_partI_forward"),
        ("yoho_partI_forward", (h, N, 8, p(eqv), N, N, N), EINVAL, "yoho_partI_forward"),
        ("yoho_partI_forward", (h, p(x), 0, p(eqv), N, N, N), EINVAL, "B=0"),
        ("yoho_partI_forward", (h, p(x), -3, p(eqv), N, N, N), EINVAL, "B=-3"),
        ("yoho_partI_forward", (h, off(x, 4), 7, p(eqv), N, N, N), EINVAL, "16-byte aligned"),
        ("yoho_partI_forward", (h, p(x), 8, p(eqv), off(inv, 4), N, N), EINVAL, "16-byte aligned"),
        ("yoho_partI_forward", (hb, p(x), 8, p(eqv), N, N, N), ENOWEIGHTS, "not loaded"),
        ("yoho_partI_forward_pair", (h, p(x), 4, N, 4, p(eqv), N, N, N), EINVAL, "yoho_partI_forward_pair"),
        ("yoho_partI_forward_pair", (h, p(x), 16000, p(x), 385, p(eqv), N, N, N), EINVAL, "16384"),
        ("yoho_partI_forward_pair", (hb, p(x), 4, p(x), 4, p(eqv), N, N, N), ENOWEIGHTS, "not loaded"),
        ("yoho_group_mean_np", (h, N, 8, p(inv), N), EINVAL, "yoho_group_mean_np"),
        ("yoho_group_mean_np", (h, p(eqv), -1, p(inv), N), EINVAL, "yoho_group_mean_np"),
        ("yoho_group_mean_np", (h, off(eqv, 4), 7, p(inv), N), EINVAL, "aligned"),
        ("yoho_nn_search", (h, N, 8, p(inv), 8, 32, 0, p(idx), N, N), EINVAL, "yoho_nn_search"),
        ("yoho_nn_search", (h, p(inv), 8, p(inv), 0, 32, 0, p(idx), N, N), EINVAL, "yoho_nn_search"),
        ("yoho_nn_search", (h, p(inv), 8, p(inv), 8, 7, 0, p(idx), N, N), EINVAL, "yoho_nn_search"),
        ("yoho_nn_search", (h, off(inv, 4), 7, p(inv), 8, 32, 0, p(idx), N, N), EINVAL, "aligned"),
        ("yoho_mutual_nn", (h, p(inv), 0, p(inv), 8, p(pairs), p(cnt), N), EINVAL, "yoho_mutual_nn"),
        ("yoho_mutual_nn", (h, p(inv), 8, p(inv), 8, N, p(cnt), N), EINVAL, "yoho_mutual_nn"),
        ("yoho_mutual_nn", (h, p(inv), 8, p(inv), 8, p(pairs), N, N), EINVAL, "yoho_mutual_nn"),
        ("yoho_des2r", (h, N, p(eqv), 8, p(idx), N, N), EINVAL, "yoho_des2r"),
        ("yoho_des2r", (h, p(eqv), p(eqv), -1, p(idx), N, N), EINVAL, "yoho_des2r"),
        ("yoho_des2r_indexed", (h, p(eqv), p(idx), p(eqv), p(idx), 0, 8, p(idx), N, N), EINVAL, "yoho_des2r_indexed"),
        ("yoho_des2r_indexed", (h, p(eqv), N, p(eqv), p(idx), 1, 8, p(idx), N, N), EINVAL, "yoho_des2r_indexed"),
        ("yoho_partII_forward", (h, p(eqv), p(eqv), p(eqv), p(eqv), p(idx), -1, p(quat), N), EINVAL, "yoho_partII_forward"),
        ("yoho_partII_forward", (h, p(eqv), N, p(eqv), p(eqv), p(idx), 8, p(quat), N), EINVAL, "yoho_partII_forward"),
        ("yoho_partII_forward", (h, p(eqv), p(eqv), p(eqv), p(eqv), p(idx), 8, N, N), EINVAL, "yoho_partII_forward"),
        ("yoho_partII_forward", (hb, p(eqv), p(eqv), p(eqv), p(eqv), p(idx), 8, p(quat), N), ENOWEIGHTS, "not loaded"),
        ("yoho_partII_forward_indexed", (h, p(eqv), p(idx), p(eqv), p(idx), p(eqv), p(idx), p(eqv), p(idx), 0, p(idx), 8, p(quat), N), EINVAL, "yoho_partII_forward_indexed"),
        ("yoho_partII_forward_indexed", (h, p(eqv), p(idx), p(eqv), N, p(eqv), p(idx), p(eqv), p(idx), 1, p(idx), 8, p(quat), N), EINVAL, "yoho_partII_forward_indexed"),
        ("yoho_hyp_from_quat", (h, p(quat), p(idx), p(k0), N, 8, p(T), N), EINVAL, "yoho_hyp_from_quat"),
        ("yoho_hyp_from_quat", (h, p(quat), p(idx), p(k0), p(k1), -2, p(T), N), EINVAL, "yoho_hyp_from_quat"),
        ("yoho_hyp_from_quat", (h, p(quat), p(idx), off(k0, 4), p(k1), 7, p(T), N), EINVAL, "8-byte aligned"),
        ("yoho_o_score", (h, p(k0), p(k1), 0, p(T), N, 8, 0.09, p(cnt), off(cnt, 4), N, N), EINVAL, "yoho_o_score"),
        ("yoho_o_score", (h, p(k0), p(k1), 8, p(T), N, 0, 0.09, p(cnt), off(cnt, 4), N, N), EINVAL, "yoho_o_score"),
        ("yoho_o_score", (h, p(k0), p(k1), 8, N, N, 8, 0.09, p(cnt), off(cnt, 4), N, N), EINVAL, "yoho_o_score"),
        ("yoho_c_ransac", (h, p(k0), p(k1), 8, N, N, 4, 0.07, p(T), p(cnt), off(cnt, 4), N, N, N), EINVAL, "yoho_c_ransac"),
        ("yoho_c_ransac", (h, p(k0), p(k1), 8, p(idx), N, 0, 0.07, p(T), p(cnt), off(cnt, 4), N, N, N), EINVAL, "yoho_c_ransac"),
        ("yoho_c_ransac_device", (h, p(k0), N, p(k1), N, 1, p(idx), 8, 0, 1, 0.07, p(T), p(cnt), off(cnt, 4), N, N), EINVAL, "yoho_c_ransac_device"),
        ("yoho_c_ransac_device", (h, p(k0), N, p(k1), N, 0, p(idx), 8, 10, 1, 0.07, p(T), p(cnt), off(cnt, 4), N, N), EINVAL, "yoho_c_ransac_device"),
        ("yoho_c_ransac_device", (h, p(k0), N, p(k1), N, 1, N, 8, 10, 1, 0.07, p(T), p(cnt), off(cnt, 4), N, N), EINVAL, "yoho_c_ransac_device"),
        ("yoho_group_gather", (h, p(k0), 8, p(x), p(inv), 8, 60, hostR, p(eqv), N, N), EINVAL, "yoho_group_gather"),
        ("yoho_group_gather", (h, p(k0), 8, p(x), p(inv), 8, 0, N, p(eqv), N, N), EINVAL, "yoho_group_gather"),
        ("yoho_group_scatter", (h, p(inv), 8, p(idx), 8, -1, p(eqv), N), EINVAL, "yoho_group_scatter"),
        ("yoho_group_scatter", (h, p(inv), 8, N, 8, 3, p(eqv), N), EINVAL, "yoho_group_scatter"),
        ("yoho_gconv_layer", (h, p(x), 8, 0, 32, p(x), N, 0, p(eqv), N), EINVAL, "yoho_gconv_layer"),
        ("yoho_gconv_layer", (h, p(x), 8, 32, 32, N, N, 0, p(eqv), N), EINVAL, "yoho_gconv_layer"),
        ("yoho_gconv_wgrad", (h, p(x), N, 8, 32, 32, p(eqv), N, N), EINVAL, "yoho_gconv_wgrad"),
        ("yoho_bn_stats", (h, p(x), 0, 32, p(inv), p(inv), N), EINVAL, "yoho_bn_stats"),
        ("yoho_bn_relu_apply", (h, p(x), 8, 32, N, p(inv), p(eqv), N), EINVAL, "yoho_bn_relu_apply"),
        ("yoho_bn_relu_backward", (h, p(x), p(x), p(x), 8, 32, p(inv), p(inv), p(inv), 1, N, p(inv), p(inv), N), EINVAL, "yoho_bn_relu_backward"),
        ("yoho_load_fcgf", (h, N, N, 0), EINVAL, "yoho_load_fcgf"),
        ("yoho_load_fcgf", (h, C.byref(cfg), C.cast(ptr_arr, C.POINTER(C.c_void_p)), 1), EINVAL, "yoho_load_fcgf"),
        ("yoho_fcgf_voxelize", (h, p(k0), 8, 0.0, p(idx), p(i32(8, 3)), C.byref(one), N), EINVAL, "yoho_fcgf_voxelize"),
        ("yoho_fcgf_voxelize", (h, p(k0), -1, 0.025, p(idx), p(i32(8, 3)), C.byref(one), N), EINVAL, "yoho_fcgf_voxelize"),
        ("yoho_fcgf_voxelize", (h, p(k0), 8, 0.025, p(idx), p(i32(8, 3)), N, N), EINVAL, "yoho_fcgf_voxelize"),
        ("yoho_fcgf_voxelize_rotated", (h, p(k0), 8, N, 0.025, p(idx), p(i32(8, 3)), N, C.byref(one), N), EINVAL, "yoho_fcgf_voxelize_rotated"),
        ("yoho_fcgf_voxelize_rotated_batch", (h, p(k0), 8, hostR, 0, 0.025, p(idx), p(i32(8, 3)), N, C.byref(one), N), EINVAL, "yoho_fcgf_voxelize_rotated_batch"),
        ("yoho_fcgf_voxelize_rotated_batch", (h, p(k0), 8, hostR, 65, 0.025, p(idx), p(i32(8, 3)), N, C.byref(one), N), EINVAL, "64"),
        ("yoho_rotate_select", (h, p(k0), hostR, N, 8, p(x), N), EINVAL, "yoho_rotate_select"),
        ("yoho_rotate_select", (h, p(k0), hostR, p(idx), -1, p(x), N), EINVAL, "yoho_rotate_select"),
        ("yoho_fcgf_forward", (h, N, 8, p(inv), N), EINVAL, "yoho_fcgf_forward"),
        ("yoho_fcgf_forward", (h, p(i32(8, 3)), -1, p(inv), N), EINVAL, "yoho_fcgf_forward"),
        ("yoho_fcgf_forward", (hb, p(i32(8, 3)), 8, p(inv), N), ENOWEIGHTS, "not loaded"),
        ("yoho_fcgf_forward_batch", (h, p(i32(8, 3)), offs2, 0, p(inv), N), EINVAL, "yoho_fcgf_forward_batch"),
        ("yoho_fcgf_forward_batch", (h, p(i32(8, 3)), offs2, 65, p(inv), N), EINVAL, "64"),
        ("yoho_fcgf_forward_batch", (h, p(i32(8, 3)), offs_bad, 2, p(inv), N), EINVAL, "non-decreasing"),
        ("yoho_fcgf_forward_batch", (h, p(i32(8, 3)), N, 1, p(inv), N), EINVAL, "yoho_fcgf_forward_batch"),
        ("yoho_group_transfer_batch", (h, p(k0), p(idx), 8, hostR, 0, ptr_arr, ptr_arr, m_arr, 0, p(eqv), p(x), p(idx), N), EINVAL, "yoho_group_transfer_batch"),
        ("yoho_group_transfer_batch", (h, p(k0), p(idx), 8, hostR, 1, ptr_arr, ptr_arr, m_arr, 60, p(eqv), p(x), p(idx), N), EINVAL, "60"),
        ("yoho_group_transfer_batch", (h, p(k0), p(idx), 8, hostR, 1, N, ptr_arr, m_arr, 0, p(eqv), p(x), p(idx), N), EINVAL, "yoho_group_transfer_batch"),
        ("yoho_register_pair", (h, p(x), p(x), p(eqv), p(eqv), p(inv), p(inv), p(k0), p(k1), 8, 8, 0, 0, 0.09, 1, 1, C.byref(res), N), EINVAL, "yoho_register_pair"),
        ("yoho_register_pair", (h, p(x), p(x), p(eqv), N, p(inv), p(inv), p(k0), p(k1), 8, 8, 0, 100, 0.09, 1, 1, C.byref(res), N), EINVAL, "yoho_register_pair"),
        ("yoho_register_pair", (h, p(x), p(x), p(eqv), p(eqv), p(inv), p(inv), p(k0), p(k1), 8, 8, 0, 100, 0.09, 1, 1, N, N), EINVAL, "yoho_register_pair"),
        ("yoho_vote_order", (1, -1, N), EINVAL, "yoho_vote_order"),
        ("yoho_vote_order", (1, 5, N), EINVAL, "yoho_vote_order"),
        ("yoho_c_draw_np", (N, C.byref(mtpos), hprob, hstart, hmembers, 4, htri, C.byref(one), C.byref(one)), EINVAL, "yoho_c_draw_np"),
        ("yoho_c_draw_np", (hkey, C.byref(badpos), hprob, hstart, hmembers, 4, htri, C.byref(one), C.byref(one)), EINVAL, "yoho_c_draw_np"),
        ("yoho_c_draw_np", (hkey, C.byref(mtpos), hprob, hstart_bad, hmembers, 4, htri, C.byref(one), C.byref(one)), EINVAL, "non-decreasing"),
        ("yoho_c_draw_np", (hkey, C.byref(mtpos), hprob, hstart, hmembers, -1, htri, C.byref(one), C.byref(one)), EINVAL, "yoho_c_draw_np"),
        ("yoho_set_gconv_mode", (h, 9), EINVAL, "yoho_set_gconv_mode"),
        ("yoho_set_gconv_mode", (N, 4), EINVAL, "yoho_set_gconv_mode"),
        ("yoho_set_partII_mode", (h, -1), EINVAL, "yoho_set_partII_mode"),
        ("yoho_set_partI_schedule", (h, 1024, 3), EINVAL, "yoho_set_partI_schedule"),
        ("yoho_set_nn_grid", (h, -0.5), EINVAL, "yoho_set_nn_grid"),
        ("yoho_set_nn_grid", (h, float("nan")), EINVAL, "yoho_set_nn_grid"),
        ("yoho_set_nn_prefilter", (N, 1), EINVAL, "yoho_set_nn_prefilter"),
        ("yoho_set_fcgf_sort", (N, 1, 1), EINVAL, "yoho_set_fcgf_sort"),
        ("yoho_range_status", (N, C.byref(one), N, N), EINVAL, "yoho_range_status"),
        ("yoho_set_profiling", (N, 1), EINVAL, "ctx"),
        ("yoho_get_kernel_ms", (h, 99, C.byref(ms)), EINVAL, "yoho_get_kernel_ms"),
        ("yoho_phase_profile", (N, 1), EINVAL, "ctx"),
        ("yoho_phase_read", (h, N, N, N, N), EINVAL, "yoho_phase_read"),
        ("yoho_clock_probe", (h, 0, p(i64(3)), N), EINVAL, "yoho_clock_probe"),
        ("yoho_clock_probe", (h, 20, N, N), EINVAL, "yoho_clock_probe"),
    ]
    seen = set()
    for name, args, code, text in cases:
        rc = getattr(lib, name)(*args)
        assert rc == code, (name, args, rc, err(lib))
        assert text in err(lib), (name, text, err(lib))
        seen.add(name)
    # every exported entry point (except the three that cannot fail on arguments) has at least one refusal above
    assert set(hip.SYMBOLS) - seen == {"yoho_last_error", "yoho_version", "yoho_ctx_destroy"}, set(hip.SYMBOLS) - seen
    assert lib.yoho_ctx_destroy(None) == 0                                    # destroying nothing is not an error
    torch.cuda.synchronize()                                                  # nothing was launched, nothing is pending, nothing crashed

    # the context is as it was: one real pass of every network
    out = ctx.partI_forward(x, want_inv=True)
    assert torch.isfinite(out["eqv"]).all() and torch.isfinite(out["inv"]).all()


def test_wrong_mode_for_pair_and_indexed_entries(env, hip):
    """yoho_partI_forward_pair needs the default PartI arithmetic, yoho_partII_forward_indexed / yoho_register_pair (YOHO-O) the default
    PartII arithmetic: YOHO_EINVAL with a message that says so, and the plain entries still run in the other mode."""
    lib, ctx, _ = env
    h = ctx._h
    x = torch.from_numpy(synth.unit_features(8, seed=2)).cuda()
    eqv = torch.zeros(16, 32, 60, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    idx = torch.zeros(8, dtype=torch.int64, device="cuda")
    quat = torch.zeros(8, 4, device="cuda")
    inv = torch.zeros(8, 32, device="cuda")
    k = torch.zeros(8, 3, dtype=torch.float64, device="cuda")
    res = hip.PairResultC()
    try:
        ctx.set_gconv_mode("bf16x3")
        assert lib.yoho_partI_forward_pair(h, p(x), 8, p(x), 8, p(eqv), None, None, None) == EINVAL and "default arithmetic mode" in err(lib)
        assert lib.yoho_partI_forward(h, p(x), 8, p(eqv), None, None, None) == 0
        ctx.set_partII_mode("bf16x3")
        e8 = eqv[:8]
        assert lib.yoho_partII_forward_indexed(h, p(e8), p(idx), p(e8), p(idx), p(e8), p(idx), p(e8), p(idx), 1, p(idx), 8, p(quat), None) == EINVAL
        assert "default PartII mode" in err(lib)
        assert lib.yoho_register_pair(h, p(x), p(x), p(e8), p(e8), p(inv), p(inv), p(k), p(k), 8, 8, 0, 100, 0.09, 1, 1, C.byref(res), None) == EINVAL
        assert "fp16x2 PartII arithmetic modes" in err(lib)
        for pm in ("cgemm", "cgemm8"):                        # the round-6 modes keep the Fourier first layer: both entries take them
            ctx.set_partII_mode(pm)
            assert lib.yoho_partII_forward_indexed(h, p(e8), p(idx), p(e8), p(idx), p(e8), p(idx), p(e8), p(idx), 1, p(idx), 8, p(quat), None) == 0
        ctx.set_partII_mode("bf16x3")
        assert lib.yoho_partII_forward(h, p(e8), p(e8), p(e8), p(e8), p(idx), 8, p(quat), None) == 0
        torch.cuda.synchronize()
    finally:
        ctx.set_gconv_mode("fgemm")
        ctx.set_partII_mode("fp16x2")


def test_empty_inputs_are_valid_and_launch_nothing(env):
    """B = 0 / M = 0 / K = 0 / n = 0 where the header allows it: return code 0, NULL data pointers accepted, outputs untouched."""
    lib, ctx, _ = env
    h = ctx._h
    one = C.c_int(7)
    sentinel = torch.full((4, 4), 3.0, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.yoho_group_mean_np(h, p(sentinel), 0, p(sentinel), None) == 0
    assert lib.yoho_nn_search(h, p(sentinel), 0, p(sentinel), 1, 32, 0, p(sentinel), None, None) == 0
    assert lib.yoho_des2r(h, p(sentinel), p(sentinel), 0, p(sentinel), None, None) == 0
    assert lib.yoho_partII_forward(h, None, None, None, None, None, 0, None, None) == 0
    assert lib.yoho_partII_forward_indexed(h, None, None, None, None, None, None, None, None, 2, None, 0, None, None) == 0
    assert lib.yoho_hyp_from_quat(h, p(sentinel), p(sentinel), p(sentinel), p(sentinel), 0, p(sentinel), None) == 0
    assert lib.yoho_group_scatter(h, None, 5, None, 0, 3, None, None) == 0
    assert lib.yoho_gconv_layer(h, None, 0, 32, 32, None, None, 0, None, None) == 0
    assert lib.yoho_fcgf_voxelize(h, None, 0, 0.025, None, None, C.byref(one), None) == 0 and one.value == 0
    assert lib.yoho_fcgf_forward(h, None, 0, None, None) == 0
    assert lib.yoho_fcgf_forward_batch(h, None, (C.c_int32 * 3)(0, 0, 0), 2, None, None) == 0
    assert lib.yoho_rotate_select(h, None, None, None, 0, None, None) == 0
    assert lib.yoho_vote_order(5, 0, None) == 0
    torch.cuda.synchronize()
    assert bool((sentinel == 3.0).all())


def test_env_switches_are_read_once_per_context(hip, sd2, monkeypatch):
    """Round 5: the library reads the environment in yoho_ctx_create only.  A context created under YOHO_PARTII_TAIL=staged keeps its
    staged tail after the variable is gone, one created without it is unaffected by setting the variable afterwards - and the two tails
    agree to the last bits of an fp32 sum (the staged one adds cone1's K in one chain)."""
    x = [torch.from_numpy(synth.unit_features(64, seed=s)).cuda() for s in (1, 2, 3, 4)]
    idx = torch.arange(64, device="cuda") % 60
    monkeypatch.setenv("YOHO_PARTII_TAIL", "staged")
    staged = hip.Context()
    monkeypatch.delenv("YOHO_PARTII_TAIL")
    fused = hip.Context()
    for c in (staged, fused):
        c.load_partII(sd2)
    q_fused = fused.partII_forward(*x, idx)
    q_staged = staged.partII_forward(*x, idx)
    monkeypatch.setenv("YOHO_PARTII_TAIL", "staged")             # too late for both: nothing reads it any more
    assert torch.equal(fused.partII_forward(*x, idx), q_fused) and torch.equal(staged.partII_forward(*x, idx), q_staged)
    assert not torch.equal(q_fused, q_staged) and (q_fused - q_staged).abs().max().item() < 1e-5
