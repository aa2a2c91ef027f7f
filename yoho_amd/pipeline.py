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
    __slots__ = ("match", "dr_index", "quat", "trans_pre", "best_h", "best_count", "trans", "order", "eqv", "range_repeats", "hyp_rows", "matches")


def describe_pair(ctx, feat0, feat1, check_range=True):
    """PartI on both fragments in one descriptor batch (tests/extractor.py:51-59 batches keypoints the same way; larger
    launches fill the chip better than two half-size passes) -> two dicts {eqv, inv_np}."""
    n0 = feat0.shape[0]
    if ctx.supports_pair(n0 + feat1.shape[0]):
        o = ctx.partI_forward_pair(feat0, feat1, want_inv=False, want_inv_np=True, check_range=check_range)
    else:                                                  # direct-conv modes or more than one pass: the caller concatenates
        o = ctx.partI_forward(torch.cat([feat0, feat1]), want_inv=False, want_inv_np=True, check_range=check_range)
    return {k: v[:n0] for k, v in o.items()}, {k: v[n0:] for k, v in o.items()}


def run_pair(ctx, feat0, feat1, keys0, keys1, inlier_dist=0.09, max_iter=1000, order_rng=None, eqv=None, estimator="yohoo", seed=0,
             hypotheses="all"):
    """feat0/feat1 (K,32,60) f32 cuda (FCGF group features), keys0/keys1 (K,3) f64 cuda.
    estimator 'yohoo' (tests/evaluator.py:112-117: PartII + one-shot vote over <= max_iter per-match hypotheses, order
    shuffled by order_rng) or 'yohoc' (tests/evaluator.py:41-47: max_iter Kabsch RANSAC iterations sampled on the device
    from the Philox stream `seed`; no PartII).  Returns PairResult with device tensors (trans is a (3,4) f64 host array,
    eye(4) rows if no hypothesis has an inlier, as tests/estimator.py:327-336; best_h is the reference's recalltime).
    hypotheses (YOHO-O): "all" = PartII and [R|t] for every match, as the reference's stage PartII_R_pre leaves them in
    Match/Trans_pre (tests/extractor.py:142-201) - quat / trans_pre have M rows; "selected" = only for the min(max_iter, M) matches
    the vote will actually read (tests/estimator.py:321-326: T = Trans[index[:max_iter]]) - quat / trans_pre then have H rows in
    vote order, hyp_rows holds their match rows; winner, count and trans are identical (a match's PartII output does not depend on
    which other matches share its pass), the other M - H hypotheses - which nothing downstream of the vote reads - are not computed."""
    r = PairResult()
    r.range_repeats = 0
    r.quat = r.trans_pre = r.order = r.hyp_rows = None
    r.matches = 0
    if hypotheses not in ("all", "selected"):
        raise ValueError(f"hypotheses must be 'all' or 'selected', got {hypotheses!r}")
    if eqv is None:
        o0, o1 = describe_pair(ctx, feat0, feat1, check_range=False)
    else:
        o0, o1 = eqv
    # tests/matcher.py:35-48 (the match count is read back here: first host wait of the pair)
    match = ctx.mutual_nn(o0["inv_np"], o1["inv_np"])
    if eqv is None and ctx.partI_overflow():
        r.range_repeats += 1
        wide = ctx._repeat_wider("gconv", lambda: describe_pair(ctx, feat0, feat1, check_range=False))
        o0, o1 = wide
        match = ctx.mutual_nn(o0["inv_np"], o1["inv_np"])
    r.eqv = (o0, o1)
    r.match = match
    M = r.matches = match.shape[0]
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

    selected = hypotheses == "selected" and H < M
    if selected:
        sel_d = order_d[:H].contiguous()
        match_s, dr_s = match[sel_d].contiguous(), r.dr_index[sel_d].contiguous()
        k0s, k1s = k0m[sel_d].contiguous(), k1m[sel_d].contiguous()
        vote_order = torch.arange(H, dtype=torch.int64, device=feat0.device)
        r.hyp_rows = sel_d

    def head_and_vote():
        # tests/extractor.py:125-138 batch_create (0<->1 exchange) + utils/network.py:259-278
        if selected:
            if ctx.supports_matched():
                r.quat = ctx.partII_forward_matched(feat0, feat1, o0["eqv"], o1["eqv"], match_s, dr_s, check_range=False)
            else:                                          # the bf16x3 repeat of the range guard takes gathered rows
                s0, s1 = match_s[:, 0], match_s[:, 1]
                r.quat = ctx.partII_forward(feat1[s1], feat0[s0], o1["eqv"][s1], o0["eqv"][s0], dr_s, check_range=False)
            r.trans_pre = ctx.hyp_from_quat(r.quat, dr_s, k0s, k1s)
            res, _ = ctx.o_score(k0m, k1m, r.trans_pre, vote_order, H, inlier_dist)
            return res.cpu().numpy()
        if ctx.supports_matched():
            r.quat = ctx.partII_forward_matched(feat0, feat1, o0["eqv"], o1["eqv"], match, r.dr_index, check_range=False)
        else:                                              # other PartII arithmetic modes take gathered rows
            r.quat = ctx.partII_forward(feat1[m1], feat0[m0], o1["eqv"][m1], o0["eqv"][m0], r.dr_index, check_range=False)
        r.trans_pre = ctx.hyp_from_quat(r.quat, r.dr_index, k0m, k1m)
        res, _ = ctx.o_score(k0m, k1m, r.trans_pre, order_d, H, inlier_dist)      # tests/estimator.py:321-336
        return res.cpu().numpy()                           # second host wait: the winner

    bh, bc = (int(v) for v in head_and_vote())
    if ctx.partII_overflow():
        r.range_repeats += 1
        bh, bc = (int(v) for v in ctx._repeat_wider("partII", head_and_vote))
    r.best_h, r.best_count = bh, bc
    r.trans = r.trans_pre[bh if selected else int(order[bh])].cpu().numpy() if bc > 0 else np.eye(4)
    return r


class PairStreamer:
    """Throughput mode for many pairs: two pairs in flight on two HIP streams.

    The descriptor pass of a pair keeps the matrix pipes busy for ~5 ms and ends in a host read-back (the match count);
    the rest of the pair (matcher, coarse rotation, PartII / Kabsch, vote: ~1.4 ms) is a chain of short kernels with two
    more read-backs.  Run back to back, the chip idles through every read-back and the short kernels never fill it.  Here
    the descriptor pass of pair i+1 is queued on stream A BEFORE the host starts waiting on pair i's read-backs on stream
    B, so both streams always have work: per pair the time of the descriptor pass remains, the rest hides behind it.

    Three library contexts (a context owns one workspace and one pair of range flags, and is single-stream by contract,
    include/yoho_hip.h): two descriptor contexts used alternately - the flags of pair i are read while pair i+1 runs in the
    other one - and one for the estimator side.  Results are those of run_pair, pair by pair (tests/test_gpu_fullsize.py).
    """

    def __init__(self, make_context, sd_partI, sd_partII=None):
        """make_context() -> a fresh hip.Context on the current device; the state dicts are loaded into the three contexts"""
        self.desc = [make_context(), make_context()]
        for c in self.desc:
            c.load_partI(sd_partI)
        self.est = make_context()
        if sd_partII is not None:
            self.est.load_partII(sd_partII)
        # the estimator side gets the high-priority stream: its short kernels must not queue behind the thousands of workgroups of
        # a descriptor launch, they slot in as soon as workgroups retire
        self.sa, self.sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)

    def set_modes(self, gconv=None, partII=None):
        for c in self.desc:
            if gconv:
                c.set_gconv_mode(gconv)
        if partII:
            self.est.set_partII_mode(partII)

    def _describe(self, i, pair):
        f0, f1 = pair[0], pair[1]
        with torch.cuda.stream(self.sa):
            o0, o1 = describe_pair(self.desc[i & 1], f0, f1, check_range=False)
            ev = torch.cuda.Event()
            ev.record(self.sa)
        for o in (o0, o1):
            for t in o.values():
                t.record_stream(self.sb)
        return o0, o1, ev

    def _register(self, pair, o0, o1, inlier_dist, max_iter, estimator, seed, hypotheses):
        """the estimator side of one pair as ONE library call (yoho_register_pair, csrc/pair.hip: mutual NN -> Des2R -> PartII +
        [R|t] + vote | device-sampled YOHO-C; vote order = numpy's RandomState(seed & 0xFFFFFFFF).shuffle restated in C): no
        tensor-library kernel runs between the descriptor pass and the winner.  The result carries what the one call returns
        (winner, count, transform, match count); match list / coarse rotations / quaternions stay in the library's scratch.
        A PartII range flag repeats the pair through the staged entries in bf16x3, as run_dataset does."""
        f = self.est.register_pair(pair[0], pair[1], o0["eqv"], o1["eqv"], o0["inv_np"], o1["inv_np"], pair[2], pair[3], estimator=estimator,
                                   max_iter=max_iter, inlier_dist=inlier_dist, seed=seed, selected=(hypotheses == "selected"))
        if f["range_flag"]:
            r = self.est._repeat_wider("partII", lambda: run_pair(
                self.est, pair[0], pair[1], pair[2], pair[3], inlier_dist=inlier_dist, max_iter=max_iter,
                order_rng=np.random.RandomState(seed & 0xFFFFFFFF), eqv=(o0, o1), estimator=estimator, seed=seed, hypotheses=hypotheses))
            r.range_repeats += 1
            return r
        r = PairResult()
        r.match = r.dr_index = r.quat = r.trans_pre = r.order = r.hyp_rows = None
        r.eqv = (o0, o1)
        r.range_repeats = 0
        r.matches, r.best_h, r.best_count, r.trans = f["matches"], f["best_h"], f["best_count"], f["trans"]
        return r

    def run(self, pairs, inlier_dist=0.09, max_iter=1000, order_rng=None, estimator="yohoo", seeds=None, hypotheses="all", keep="all",
            fused=False):
        """pairs: sequence of (feat0, feat1, keys0, keys1) device tensors.  Returns the list of PairResult; keep="last": only the
        last pair's (a PairResult holds its descriptors, 77 MB at 2 x 5000 keypoints: a long throughput run must not keep them all).
        fused: the estimator side of every pair is one library call (_register; needs `seeds`, which then also seed the YOHO-O vote
        order - order_rng is not consumed); same winner and transform as the staged composition with
        order_rng = RandomState(seed & 0xFFFFFFFF) (tests/test_gpu_fullsize.py)."""
        if fused and seeds is None:
            raise ValueError("PairStreamer.run(fused=True) takes the vote order / sampling stream of every pair from `seeds`")
        fused = fused and (estimator == "yohoc" or self.est.supports_matched())
        pairs = list(pairs)
        out = []
        if not pairs:
            return out
        cur = torch.cuda.current_stream()
        self.sa.wait_stream(cur)
        self.sb.wait_stream(cur)
        nxt = self._describe(0, pairs[0])
        for i, pair in enumerate(pairs):
            o0, o1, ev = nxt
            if i + 1 < len(pairs):
                nxt = self._describe(i + 1, pairs[i + 1])           # queued before this pair's read-backs block the host
            with torch.cuda.stream(self.sb):
                self.sb.wait_event(ev)
                d = self.desc[i & 1]
                repeats = 0
                # the descriptor pass's range flag BEFORE the pair runs, read on this stream (it waits for the pass, as the match
                # count's read-back would a moment later; the next pair's pass is already queued on the other stream): a pass that
                # left the fp16 range is repeated in bf16x3 first, so the pair runs once and consumes the RNG once, as run_pair does
                if d.partI_overflow():
                    o0, o1 = d._repeat_wider("gconv", lambda: describe_pair(d, pair[0], pair[1], check_range=False))
                    torch.cuda.current_stream().synchronize()        # d's next pass is queued on the other stream
                    repeats = 1
                if fused:
                    r = self._register(pair, o0, o1, inlier_dist, max_iter, estimator, seeds[i], hypotheses)
                else:
                    r = run_pair(self.est, pair[0], pair[1], pair[2], pair[3], inlier_dist=inlier_dist, max_iter=max_iter, order_rng=order_rng,
                                 eqv=(o0, o1), estimator=estimator, seed=(seeds[i] if seeds is not None else 0), hypotheses=hypotheses)
                r.range_repeats += repeats
            if keep == "last":
                out = [r]
            else:
                out.append(r)
        cur.wait_stream(self.sa)
        cur.wait_stream(self.sb)
        return out
