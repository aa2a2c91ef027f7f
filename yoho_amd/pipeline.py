"""HBM-resident pair pipeline: descriptor (PartI) -> invariant pooling -> mutual NN -> coarse
rotation index -> PartII -> per-match hypotheses -> YOHO-O vote, with no disk round trips.

This is the fast path behind the drop-in classes (extractor / matcher / estimator keep the
reference's .npy stage-cache interface); bench.py and smoke() time exactly this function.
It composes the same C-ABI calls in the order tests/evaluator.py:112-117 runs the stages.

fp16 range guard: the default arithmetic keeps activations in fp16 planes; the kernels raise a device flag
when a value does not fit (include/yoho_hip.h, yoho_range_status).  The stages are launched without
waiting for that flag; it is read where the host has to wait anyway (the match count, the winner), and a
raised flag repeats the pair with the affected network in bf16x3 planes.
"""
import numpy as np
import torch


class PairResult:
    __slots__ = ("match", "dr_index", "quat", "trans_pre", "best_h", "best_count", "trans", "order", "eqv", "range_repeats")


def describe_pair(ctx, feat0, feat1, check_range=True):
    """PartI on both fragments in one descriptor batch (tests/extractor.py:51-59 batches keypoints the same way; larger
    launches fill the chip better than two half-size passes) -> two dicts {eqv, inv_np}."""
    n0 = feat0.shape[0]
    if ctx.supports_pair(n0 + feat1.shape[0]):
        o = ctx.partI_forward_pair(feat0, feat1, want_inv=False, want_inv_np=True, check_range=check_range)
    else:                                                  # direct-conv modes or more than one pass: the caller concatenates
        o = ctx.partI_forward(torch.cat([feat0, feat1]), want_inv=False, want_inv_np=True, check_range=check_range)
    return {k: v[:n0] for k, v in o.items()}, {k: v[n0:] for k, v in o.items()}


def run_pair(ctx, feat0, feat1, keys0, keys1, inlier_dist=0.09, max_iter=1000, order_rng=None, eqv=None, estimator="yohoo", seed=0):
    """feat0/feat1 (K,32,60) f32 cuda (FCGF group features), keys0/keys1 (K,3) f64 cuda.
    estimator 'yohoo' (tests/evaluator.py:112-117: PartII + one-shot vote over <= max_iter per-match hypotheses, order
    shuffled by order_rng) or 'yohoc' (tests/evaluator.py:41-47: max_iter Kabsch RANSAC iterations sampled on the device
    from the Philox stream `seed`; no PartII).  Returns PairResult with device tensors (trans is a (3,4) f64 host array,
    eye(4) rows if no hypothesis has an inlier, as tests/estimator.py:327-336; best_h is the reference's recalltime)."""
    r = PairResult()
    r.range_repeats = 0
    r.quat = r.trans_pre = r.order = None
    if eqv is None:
        o0, o1 = describe_pair(ctx, feat0, feat1, check_range=False)
    else:
        o0, o1 = eqv
    # tests/matcher.py:35-48 (the match count is read back here: first host wait of the pair)
    match = ctx.mutual_nn(o0["inv_np"], o1["inv_np"])
    if eqv is None and ctx.range_status()[0]:
        r.range_repeats += 1
        wide = ctx._repeat_wider("gconv", lambda: describe_pair(ctx, feat0, feat1, check_range=False))
        o0, o1 = wide
        match = ctx.mutual_nn(o0["inv_np"], o1["inv_np"])
    r.eqv = (o0, o1)
    r.match = match
    M = match.shape[0]
    if M == 0:
        r.dr_index = r.quat = r.trans_pre = None
        r.best_h, r.best_count, r.trans, r.order = 0, 0, np.eye(4), None
        return r
    m0, m1 = match[:, 0], match[:, 1]
    # tests/extractor.py:97-99: Batch_Des2R_torch(feats1, feats0); rows addressed in place through the match list
    r.dr_index = ctx.des2r_matched(o1["eqv"], o0["eqv"], match)
    if estimator == "yohoc":
        # tests/estimator.py:28-141 with statistic, sampling, Kabsch and vote on the device; one read-back
        T, res, _ = ctx.c_ransac_device(keys0, keys1, r.dr_index, max_iter, seed, inlier_dist, match=match)
        host = torch.cat([T.reshape(-1), res.to(torch.float64)]).cpu().numpy()
        r.best_h, r.best_count = int(host[12]), int(host[13])
        r.trans = host[:12].reshape(3, 4) if r.best_count > 0 else np.eye(4)
        return r
    if estimator != "yohoo":
        raise ValueError(f"estimator must be 'yohoo' or 'yohoc', got {estimator!r}")
    k0m, k1m = keys0[m0].contiguous(), keys1[m1].contiguous()
    order = np.arange(M)
    (order_rng if order_rng is not None else np.random).shuffle(order)      # tests/estimator.py:321-323
    r.order = order
    order_d = torch.from_numpy(order).to(feat0.device)
    H = min(max_iter, M)

    def head_and_vote():
        # tests/extractor.py:125-138 batch_create (0<->1 exchange) + utils/network.py:259-278
        if ctx.supports_matched():
            r.quat = ctx.partII_forward_matched(feat0, feat1, o0["eqv"], o1["eqv"], match, r.dr_index, check_range=False)
        else:                                              # other PartII arithmetic modes take gathered rows
            r.quat = ctx.partII_forward(feat1[m1], feat0[m0], o1["eqv"][m1], o0["eqv"][m0], r.dr_index, check_range=False)
        r.trans_pre = ctx.hyp_from_quat(r.quat, r.dr_index, k0m, k1m)
        res, _ = ctx.o_score(k0m, k1m, r.trans_pre, order_d, H, inlier_dist)      # tests/estimator.py:321-336
        return res.cpu().numpy()                           # second host wait: the winner

    bh, bc = (int(v) for v in head_and_vote())
    if ctx.range_status()[1]:
        r.range_repeats += 1
        bh, bc = (int(v) for v in ctx._repeat_wider("partII", head_and_vote))
    r.best_h, r.best_count = bh, bc
    r.trans = r.trans_pre[int(order[bh])].cpu().numpy() if bc > 0 else np.eye(4)
    return r
