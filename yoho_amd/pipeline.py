"""HBM-resident pair pipeline: descriptor (PartI) -> invariant pooling -> mutual NN -> coarse
rotation index -> PartII -> per-match hypotheses -> YOHO-O vote, with no disk round trips.

This is the fast path behind the drop-in classes (extractor / matcher / estimator keep the
reference's .npy stage-cache interface); bench.py and smoke() time exactly this function.
It composes the same C-ABI calls in the order tests/evaluator.py:112-117 runs the stages.
"""
import numpy as np
import torch


class PairResult:
    __slots__ = ("match", "dr_index", "quat", "trans_pre", "best_h", "best_count", "trans", "order")


def run_pair(ctx, feat0, feat1, keys0, keys1, inlier_dist=0.09, max_iter=1000, order_rng=None, eqv=None):
    """feat0/feat1 (K,32,60) f32 cuda (FCGF group features), keys0/keys1 (K,3) f64 cuda.
    Returns PairResult with device tensors (trans is a (3,4) f64 host array, eye(4) rows if no
    hypothesis has an inlier, as tests/estimator.py:327-336)."""
    if eqv is None:
        # both fragments in one descriptor batch (tests/extractor.py:51-59 batches keypoints the same way): larger
        # launches fill the chip better than two half-size passes
        n0 = feat0.shape[0]
        try:
            o = ctx.partI_forward_pair(feat0, feat1, want_inv=False, want_inv_np=True)
        except RuntimeError:                               # non-default arithmetic mode (or > 16384 keypoints)
            o = ctx.partI_forward(torch.cat([feat0, feat1]), want_inv=False, want_inv_np=True)
        o0 = {k: (v[:n0] if v is not None else None) for k, v in o.items()}
        o1 = {k: (v[n0:] if v is not None else None) for k, v in o.items()}
    else:
        o0, o1 = eqv
    # tests/matcher.py:35-48
    match = ctx.mutual_nn(o0["inv_np"], o1["inv_np"])
    r = PairResult()
    r.match = match
    M = match.shape[0]
    if M == 0:
        r.dr_index = r.quat = r.trans_pre = None
        r.best_h, r.best_count, r.trans, r.order = 0, 0, np.eye(4), None
        return r
    m0, m1 = match[:, 0], match[:, 1]
    # tests/extractor.py:97-99: Batch_Des2R_torch(feats1, feats0); rows addressed in place through the match list
    r.dr_index = ctx.des2r_matched(o1["eqv"], o0["eqv"], match)
    # tests/extractor.py:125-138 batch_create (0<->1 exchange) + utils/network.py:259-278
    try:
        r.quat = ctx.partII_forward_matched(feat0, feat1, o0["eqv"], o1["eqv"], match, r.dr_index)
    except RuntimeError:                                   # non-default PartII arithmetic mode: gather first
        r.quat = ctx.partII_forward(feat1[m1], feat0[m0], o1["eqv"][m1], o0["eqv"][m0], r.dr_index)
    k0m, k1m = keys0[m0].contiguous(), keys1[m1].contiguous()
    r.trans_pre = ctx.hyp_from_quat(r.quat, r.dr_index, k0m, k1m)
    # tests/estimator.py:321-336
    order = np.arange(M)
    (order_rng if order_rng is not None else np.random).shuffle(order)
    r.order = order
    H = min(max_iter, M)
    res, _ = ctx.o_score(k0m, k1m, r.trans_pre, torch.from_numpy(order).to(feat0.device), H, inlier_dist)
    bh, bc = (int(v) for v in res.cpu().numpy())
    r.best_h, r.best_count = bh, bc
    r.trans = r.trans_pre[int(order[bh])].cpu().numpy() if bc > 0 else np.eye(4)
    return r
