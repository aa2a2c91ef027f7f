"""Drop-in for simple_yoho/yoho_extract.py: one-cloud API  pc -> (kpts, inv, eqv).

    extractor = yoho_extractor(fcgf_ckpt, yoho_ckpt, fcgf=my_backbone)
    kpts, feat_inv, feat_eqv = extractor.run(pc, voxel_size=0.025, nkpts=5000)

The FCGF sparse-conv backbone (simple_yoho/fcgf_feat.py on MinkowskiEngine) is ``yoho_amd.fcgf_feat.fcgf_extractor``
(csrc/sparse.hip), built from ``fcgf_ckpt`` as the reference does (:20); any object with ``run(pc, voxel_size) ->
(ds_points (n,3), unit-norm feats (n,32))`` can be passed as ``fcgf`` instead.  The 60-fold NN feature transfer and
the PartI group conv run on the HIP library too.

Differences from the reference (all deliberate):
  * no argparse at import time (simple_yoho/yoho_extract.py:11-13 breaks under pytest / other argv);
  * the NN result is unpacked as (dists, inds) - the reference names them the other way round
    (:37) and would index the features with distances;
  * any number of keypoints works (no "avoid B == 1" last-batch merge needed, :55-58).
"""
import os
import threading

import numpy as np
import torch

from . import hip
from . import weights as W
from .utils import transform_points


# Page-locked result tensors are handed out only while the pinned bytes still ALIVE in the caller's hands stay under this budget
# (YOHO_PIN_OUTPUT_BYTES, default 2 GiB; 0 = always pageable, as the reference returns them).  A caller that streams fragments
# (describe, consume, drop) keeps the fast copy; one that keeps the descriptors of a whole scene or test set - 38 MB per fragment,
# the testset and cache writers do - gets ordinary pageable tensors once the budget is used up instead of pinning GBs of host RAM
# that torch's caching host allocator never gives back to the OS.
PIN_OUTPUT_BYTES = int(os.environ.get("YOHO_PIN_OUTPUT_BYTES", str(2 << 30)))
_pinned_alive = [0]
_pinned_lock = threading.Lock()


def _unpin(nbytes):
    with _pinned_lock:
        _pinned_alive[0] -= nbytes


def _to_host(*tensors):
    """device tensors -> CPU tensors, as the reference API returns them (simple_yoho/yoho_extract.py:72-77), through page-locked memory
    while the budget above allows: a pageable 38 MB copy of the descriptors took 2.5 ms of a 48 ms fragment, the same copy into a pinned
    tensor 0.8 ms.  The pinned blocks come from torch's caching host allocator (no hipHostMalloc per call after the first); every call
    gets tensors of its own."""
    import weakref
    need = sum(t.numel() * t.element_size() for t in tensors)
    with _pinned_lock:                                 # reserve the whole call's bytes or none (several extractor threads share the budget)
        pin = need > 0 and _pinned_alive[0] + need <= PIN_OUTPUT_BYTES
        if pin:
            _pinned_alive[0] += need
    outs = [torch.empty(t.shape, dtype=t.dtype, pin_memory=pin) for t in tensors]
    for o, t in zip(outs, tensors):
        o.copy_(t, non_blocking=pin)
        if pin:
            # released when the page-locked STORAGE dies, not the tensor object: out.numpy(), a view or a slice keep the storage (and
            # the pinned pages) alive after the tensor the caller was handed is gone
            weakref.finalize(o.untyped_storage(), _unpin, o.numel() * o.element_size())
    torch.cuda.current_stream().synchronize()
    return tuple(outs)


class yoho_extractor():
    def __init__(self, fcgf_ckpt='model/Backbone/best_val_checkpoint.pth', yoho_ckpt='model/PartI_train/model_best.pth',
                 fcgf=None, so3_dir=None):
        self.ctx = hip.get_context(so3_dir=so3_dir)
        self.grs = self.ctx.tables.R64
        self.fcgf_ckpt = fcgf_ckpt
        if fcgf is None and fcgf_ckpt is not None:
            from .fcgf_feat import fcgf_extractor
            fcgf = fcgf_extractor(fcgf_ckpt, ctx=self.ctx)
        self.fcgf = fcgf
        self.yoho_ckpt = yoho_ckpt
        self._load_model()
        self.bs = 500
        # rotated copies of the cloud per backbone pass (HBM-resident path; split further by voxel count): one number, or a list
        # "9,17,17,17" (the first pass's voxelisation and maps are the only ones nothing hides - a smaller first pass exposes less)
        rb = [int(v) for v in os.environ.get("YOHO_ROT_BATCH", "15").split(",")]
        self.rot_batch = rb[0] if len(rb) == 1 else rb
        self.overlap_keypoint_draw = os.environ.get("YOHO_OVERLAP_DRAW", "1") != "0"   # keypoint permutation drawn while the first backbone pass runs
        # backbone passes alternate between two lanes (stream + library context = workspace): a pass's voxelisation, coordinate and
        # kernel maps - atomics and scans with host round trips for the level sizes - are queued while the previous pass's
        # convolutions still run on the other lane (YOHO_FCGF_LANES=1: one lane, as up to round 5)
        self.lanes = max(1, min(2, int(os.environ.get("YOHO_FCGF_LANES", "2"))))
        self._main_stream, self._side_stream, self._side_of, self._tail = None, None, None, None
        if self.lanes > 1 and hasattr(self.fcgf, "lane_context"):
            self.fcgf.lane_context()               # the second lane's weights are resident from here on, like the first one's

    def _load_model(self):
        sd = self.yoho_ckpt if isinstance(self.yoho_ckpt, dict) else W.load_checkpoint(self.yoho_ckpt)[0]
        self._sd = W.to_numpy_state_dict(sd)
        W.check_state_dict(self._sd, W.PARTI_SPEC, strict=False)
        self.ctx.load_partI(self._sd, owner=self)

    def _partI(self, kpts_f):
        if self.ctx.partI_owner is not self:          # another network object loaded its checkpoint into the shared context
            self.ctx.load_partI(self._sd, owner=self)
        return self.ctx.partI_forward(kpts_f.contiguous(), want_inv=True)

    def _feature_transfer_xyz(self, query, source, source_f):
        """NN in xyz (fp32, 'SquareL2') and feature row transfer (simple_yoho/yoho_extract.py:33-39)."""
        q = torch.from_numpy(np.asarray(query).astype(np.float32)).cuda().contiguous()
        s = (source if isinstance(source, torch.Tensor) else torch.from_numpy(np.asarray(source))).to(device="cuda", dtype=torch.float32).contiguous()
        f = (source_f if isinstance(source_f, torch.Tensor) else torch.from_numpy(np.asarray(source_f))).to(device="cuda", dtype=torch.float32)
        dist, idx = self.ctx.nn_search(q, s, want_dist=False, squared=True)
        return f[idx]

    def _transfer(self, res, pc_d, Rs, kidx_d, g0, kpts_f, ctx=None):
        """NN feature transfer of one backbone pass: kpts_f[:, :, g0 + j] = F_j[nn(R_j keypoints, down-sampled points of copy j)]"""
        ctx = self.ctx if ctx is None else ctx
        if hasattr(ctx, "group_transfer_batch") and all(f.shape[1] == 32 for _, f, _ in res):
            # one library call for the pass (the same three kernels per copy, queued from C: no binding round trips in between)
            ctx.group_transfer_batch(pc_d, kidx_d, list(Rs), [ds for _, _, ds in res], [f.contiguous() for _, f, _ in res], g0, kpts_f)
            return
        for j, (sel, pci_f, ds) in enumerate(res):
            q = ctx.rotate_select(pc_d, Rs[j], kidx_d)
            _, idx = ctx.nn_search(q, ds, want_dist=False, squared=True)
            ctx.group_scatter(pci_f, idx, g0 + j, kpts_f)

    def _pass_starts(self, G):
        """first group element of every backbone pass, and G"""
        rb = self.rot_batch
        sizes = list(rb) if isinstance(rb, (list, tuple)) else [int(rb)] * ((G + int(rb) - 1) // int(rb))
        starts = [0]
        for n in sizes:
            if starts[-1] >= G:
                break
            starts.append(min(G, starts[-1] + max(1, int(n))))
        while starts[-1] < G:                      # a list that does not cover the group: its last size repeats
            starts.append(min(G, starts[-1] + max(1, int(sizes[-1]))))
        return starts

    def _lanes(self):
        """[(library context, torch stream)] the backbone passes alternate over: the caller's stream with the extractor's context,
        and - with two lanes - a side stream with the backbone's second context (its own workspace).  Two things are measured rather
        than assumed (hip.concurrent_stream): HIP deals a process's streams round-robin onto four hardware queues, so one candidate in
        four would run its kernels strictly behind the other lane's; and the NULL stream is never a lane - with it as lane 0 the
        second lane's map kernels were found starved for the whole length of lane 0's convolutions in some processes (bench.py without
        its dataset and sustained legs: 40.5 against 36.9 ms per fragment) - so a caller on the null stream gets a lane 0 stream of
        the extractor's own, joined to the caller's stream on both sides."""
        cur = torch.cuda.current_stream()
        if self.lanes < 2 or not hasattr(self.fcgf, "lane_context"):
            return [(self.ctx, cur)]
        if self._side_stream is None or self._side_of != cur.cuda_stream:
            main = cur if cur.cuda_stream != 0 else hip.concurrent_stream(self.ctx, [cur])
            self._main_stream, self._side_stream = main, hip.concurrent_stream(self.ctx, [cur, main] if main is not cur else [cur])
            self._side_of, self._tail = cur.cuda_stream, None
        return [(self.ctx, self._main_stream), (self.fcgf.lane_context(), self._side_stream)]

    def _queue_passes(self, pc, voxel_size, nkpts):
        """Queue a fragment's backbone passes and NN feature transfers on the lanes; returns without joining them:
        dict(kpts, kpts_f (K,32,60) cuda - complete once `done` has run -, done = one event per lane, keep = tensors that must outlive
        `done`).  The keypoint draw sits off the critical path: the reference's np.random.permutation(len(pc))[0:nkpts] on the global
        generator costs 5 ms of host time for 300 k points, and nothing on the device depends on it until the first NN transfer - it is
        taken after one backbone pass per lane has been queued (a library call returns once the last level size is known, with most of
        the pass still running on the device), so the device works while the host shuffles.  It is the only draw in this method, so the
        generator is consumed exactly as in the reference (same keypoints for the same seed).  The backbone passes alternate over the
        lanes of `_lanes()`; which lane a pass runs on changes no bit of its features."""
        G = self.grs.shape[0]
        starts = self._pass_starts(G)
        batches = [[self.grs[i] for i in range(i0, i1)] for i0, i1 in zip(starts[:-1], starts[1:])]
        lanes = self._lanes()
        main = lanes[0][1]
        cur = torch.cuda.current_stream()
        if main is not cur:
            main.wait_stream(cur)                  # lane 0 is the extractor's own stream (the caller is on the null stream)
        with torch.cuda.stream(main):
            pc_d = torch.from_numpy(np.ascontiguousarray(np.asarray(pc, dtype=np.float64))).cuda()
        uploaded = torch.cuda.Event()              # the cloud is on the device (queued on lane 0)
        uploaded.record(main)
        for _, st in lanes[1:]:
            st.wait_event(uploaded)

        def backbone(b):                           # pass b on its lane: the pass's tensors are allocated, used and released on that stream
            ctx, st = lanes[b % len(lanes)]
            with torch.cuda.stream(st):
                return self.fcgf.extract_rotated_batch(pc_d, batches[b], voxel_size, **({"ctx": ctx} if ctx is not self.ctx else {}))

        # one pass per lane is queued before the draw: the second one builds its maps while the first one's convolutions run
        ahead = [backbone(b) for b in range(min(len(lanes), len(batches)))]
        kpts_index = np.random.permutation(len(pc))[0:nkpts]
        kpts = pc[kpts_index]
        with torch.cuda.stream(main):
            kpts_f = torch.empty((kpts.shape[0], 32, 60), dtype=torch.float32, device="cuda")
            kidx_d = torch.from_numpy(kpts_index.astype(np.int64)).cuda()
        ready = torch.cuda.Event()                 # keypoint indices and the output tensor exist
        ready.record(main)
        for c, st in lanes:
            c.set_nn_grid(voxel_size)              # the NN targets are one point per voxel: grid search, same winners
            if st is not main:
                st.wait_event(ready)
        try:
            for b, Rs in enumerate(batches):
                ctx, st = lanes[b % len(lanes)]
                res = ahead[b] if b < len(ahead) else backbone(b)
                with torch.cuda.stream(st):
                    self._transfer(res, pc_d, Rs, kidx_d, starts[b], kpts_f, ctx=ctx)
                    if b < len(ahead):
                        ahead[b] = None
                    del res
        except BaseException:
            for _, st in lanes:                    # work already queued on a lane still reads / writes the tensors of this frame
                st.synchronize()
            raise
        finally:
            for c, _ in lanes:
                c.set_nn_grid(0)
        done = []
        for _, st in lanes:
            e = torch.cuda.Event()
            e.record(st)
            done.append(e)
        return {"kpts": kpts, "kpts_f": kpts_f, "done": done, "keep": (pc_d, kidx_d), "main": main}

    def _extract_features_overlapped(self, pc, voxel_size, nkpts):
        q = self._queue_passes(pc, voxel_size, nkpts)
        cur, main = torch.cuda.current_stream(), q["main"]
        with torch.cuda.stream(main):
            for e in q["done"]:
                main.wait_event(e)                 # PartI reads every column of kpts_f
            self._last_group_feats = q["kpts_f"]
            out = self._partI(q["kpts_f"])
            res = (q["kpts"],) + _to_host(out["inv"], out["eqv"])       # (waits for lane 0)
        if main is not cur:
            cur.wait_stream(main)                  # what the caller queues next sees a finished call, as on one stream
        return res

    def run_many(self, pcs, voxel_size=0.025, nkpts=5000):
        """run() over an iterable of clouds, streamed: yields (kpts, feat_inv, feat_eqv) per cloud, in order, the same values
        run() returns for the same clouds and generator state.  The descriptor pass and the result copy of fragment f run on a tail
        lane (its own stream and library context, driven by a helper thread) while the backbone lanes already work on fragment f + 1,
        so that what a single run() call leaves exposed - the first pass's voxelisation and maps, PartI, the copy of 38 MB of results -
        is hidden behind convolutions (`bench.py` fcgf leg: ms_per_fragment_streamed).  A fragment is yielded when the next one
        has been queued; at most two are in flight."""
        if self.fcgf is None or not (hasattr(self.fcgf, "extract_rotated_batch") and hasattr(self.fcgf, "lane_context")) or self.lanes < 2:
            for pc in pcs:
                yield self.run(pc, voxel_size=voxel_size, nkpts=nkpts)
            return
        lanes = self._lanes()
        if self._tail is None:
            self._tail = (hip.get_context(self.ctx.device, self.ctx.tables.dir, lane=2),
                          hip.concurrent_stream(self.ctx, [torch.cuda.current_stream()] + [st for _, st in lanes]))
        tctx, tst = self._tail

        queued = []                                # fragments whose lanes may still be running

        def finish(q):
            with torch.cuda.stream(tst):
                for e in q["done"]:
                    tst.wait_event(e)
                if tctx.partI_owner is not self:
                    tctx.load_partI(self._sd, owner=self)
                out = tctx.partI_forward(q["kpts_f"], want_inv=True)        # (its range check waits for the tail stream only)
                self._last_group_feats = q["kpts_f"]
                res = (q["kpts"],) + _to_host(out["inv"], out["eqv"])       # (waits for the tail stream: the fragment is complete)
            queued.remove(q)
            return res

        # finish(f) - PartI, its range check, the result copy: mostly waiting for the device - runs on a helper thread while this
        # one queues the passes of f + 1 (the keypoint draws stay on this thread, in order)
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1)
        try:
            prev, fut = None, None
            for pc in pcs:
                cur = self._queue_passes(pc, voxel_size, nkpts)
                queued.append(cur)
                if fut is not None:
                    yield fut.result()
                fut = pool.submit(finish, prev) if prev is not None else None
                prev = cur
            if fut is not None:
                yield fut.result()
            if prev is not None:
                yield pool.submit(finish, prev).result()
        finally:
            pool.shutdown(wait=True)
            for q in queued:                       # left early (an error, or the caller stopped iterating): queued lanes still use the
                for e in q["done"]:                # fragment's tensors - wait for them before this frame lets go of them (a finished
                    e.synchronize()                # fragment's events have long completed: no cost on the normal path)

    def extract_features(self, pc, voxel_size, nkpts=5000):
        if self.fcgf is None:
            raise NotImplementedError("no FCGF backbone: pass fcgf_ckpt=<FCGF checkpoint> or fcgf=<object with run(pc, voxel_size)>")
        if hasattr(self.fcgf, "extract_features_dev") and hasattr(self.fcgf, "extract_rotated_batch") and self.overlap_keypoint_draw:
            return self._extract_features_overlapped(pc, voxel_size, nkpts)
        kpts_index = np.random.permutation(len(pc))[0:nkpts]
        kpts = pc[kpts_index]
        kpts_f = torch.empty((kpts.shape[0], 32, 60), dtype=torch.float32, device="cuda")
        if hasattr(self.fcgf, "extract_features_dev"):
            # HBM-resident path: the cloud is uploaded once; rotation (f64), voxelisation, backbone and the NN feature
            # transfer of all 60 group elements run on the device (same operations as the loop below)
            pc_d = torch.from_numpy(np.ascontiguousarray(np.asarray(pc, dtype=np.float64))).cuda()
            kidx_d = torch.from_numpy(kpts_index.astype(np.int64)).cuda()
            G = self.grs.shape[0]
            starts = self._pass_starts(G)
            self.ctx.set_nn_grid(voxel_size)       # the NN targets are one point per voxel: grid search, same winners
            try:
                for i0, i1 in zip(starts[:-1], starts[1:]):
                    Rs = [self.grs[i] for i in range(i0, i1)]
                    if hasattr(self.fcgf, "extract_rotated_batch"):
                        # rotated copies never materialised: rotation + voxelisation + down-sampled points in one pass
                        res = self.fcgf.extract_rotated_batch(pc_d, Rs, voxel_size)
                        self._transfer(res, pc_d, Rs, kidx_d, i0, kpts_f)
                        continue
                    Rts = [torch.from_numpy(np.ascontiguousarray(R.T)).cuda() for R in Rs]
                    pcs = [pc_d @ Rt for Rt in Rts]
                    kp_d = pc_d[kidx_d]
                    for j, (pci, (sel, pci_f)) in enumerate(zip(pcs, self.fcgf.extract_features_dev_batch(pcs, voxel_size))):
                        q = (kp_d @ Rts[j]).to(torch.float32).contiguous()
                        _, idx = self.ctx.nn_search(q, pci[sel].to(torch.float32).contiguous(), want_dist=False, squared=True)
                        self.ctx.group_scatter(pci_f.contiguous(), idx, i0 + j, kpts_f)
            finally:
                self.ctx.set_nn_grid(0)
            self._last_group_feats = kpts_f
            out = self._partI(kpts_f)
            return (kpts,) + _to_host(out["inv"], out["eqv"])
        for i in range(self.grs.shape[0]):
            kptsi = transform_points(kpts.copy(), self.grs[i])
            pci = transform_points(pc.copy(), self.grs[i])
            pci_ds, pci_f = self.fcgf.run(pci, voxel_size)
            kpts_f[:, :, i] = self._feature_transfer_xyz(kptsi, pci_ds, pci_f)
        self._last_group_feats = kpts_f                     # (n,32,60) group features, kept for inspection
        out = self._partI(kpts_f)
        # output: n*32; n*32*60 (cpu tensors, as the reference)
        return (kpts,) + _to_host(out["inv"], out["eqv"])

    def run(self, pc, voxel_size=0.025, nkpts=5000):
        kpts, feat_inv, feat_eqv = self.extract_features(pc, voxel_size, nkpts=nkpts)
        return kpts, feat_inv, feat_eqv
