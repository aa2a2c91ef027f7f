"""Drop-in for YOHO_testset.py: "PC*60 rotations -> FCGF backbone -> FCGF group feature for PC keypoints".

    testset_create(cfg).batch_feature_extraction()

writes ``{output_dir}/Testset/{dataset}/{scene}/FCGF_Input_Group_feature/{pc_id}.npy`` ((K,32,60) f32, group axis in the
order of Rotation.npy) for every fragment of every scene, exactly the files tests/extractor.py:46-49 reads.

Reference flow (YOHO_testset.py:20-166), kept step for step: rotate the fragment by R_g (f64), voxelise (first point of
every voxel, :39-49), run the backbone (:143-147), rotate the keypoints by R_g and take, for each, the feature row of its
nearest down-sampled point (:153-159, KNN(1), type-promoted f64 'L2' distance).  Here the fragment is uploaded once and
all 60 group elements run on the device (yoho_fcgf_voxelize / yoho_fcgf_forward / yoho_group_gather); there is no
DataLoader batching (the reference's batch of 4 clouds only amortises MinkowskiEngine launches).

``cfg`` needs ``model`` (FCGF checkpoint path or dict), ``voxel_size``, ``dataset``; optional ``output_dir`` /
``origin_dir`` (defaults './data/YOHO_FCGF', './data/origin_data' as the reference, :61-62) and ``datasets`` (a prebuilt
{scene: dataset} dict, otherwise ``get_dataset_name(dataset, origin_dir)``).
"""
import os
import queue
import threading

import numpy as np
import torch

from . import hip
from .dataset import get_dataset_name
from .fcgf_feat import fcgf_extractor
from .utils import make_non_exists_dir


class testset_create():
    def __init__(self, config, ctx=None):
        self.config = config
        self.dataset_name = self.config.dataset
        self.output_dir = getattr(config, 'output_dir', './data/YOHO_FCGF')
        self.origin_dir = getattr(config, 'origin_dir', './data/origin_data')
        self.datasets = getattr(config, 'datasets', None) or get_dataset_name(self.dataset_name, self.origin_dir)
        self.ctx = ctx if ctx is not None else hip.get_context()
        self.Rgroup = self.ctx.tables.R64
        self.fcgf = fcgf_extractor(self.config.model, ctx=self.ctx)
        # backbone passes alternate over two lanes (stream + library context = workspace), as in yoho_extractor: a pass's voxelisation
        # and maps are queued while the previous pass's convolutions run (YOHO_FCGF_LANES=1: everything on the caller's stream).
        # Across fragments the host never waits for the device: the result goes to a page-locked buffer by an asynchronous copy and
        # is written by a writer thread while the next fragments run; the next fragment's files are read ahead by a loader thread.
        self.lanes = max(1, min(2, int(os.environ.get("YOHO_FCGF_LANES", "2"))))
        self._main_stream, self._side_stream, self._side_of = None, None, None
        self.stats = {}
        if self.lanes > 1:
            self.fcgf.lane_context()                                 # the second lane's weights are resident from here on, like the first one's

    def _lanes(self):
        """as yoho_extractor._lanes: lane streams picked by measurement (no shared hardware queue), never the null stream"""
        cur = torch.cuda.current_stream()
        if self.lanes < 2:
            return [(self.ctx, cur)]
        if self._side_stream is None or self._side_of != cur.cuda_stream:
            main = cur if cur.cuda_stream != 0 else hip.concurrent_stream(self.ctx, [cur])
            self._main_stream, self._side_stream = main, hip.concurrent_stream(self.ctx, [cur, main] if main is not cur else [cur])
            self._side_of = cur.cuda_stream
        return [(self.ctx, self._main_stream), (self.fcgf.lane_context(), self._side_stream)]

    def fragment_group_features(self, pc, keys, join=True):
        """pc (N,3), keys (K,3) f64 -> (K,32,60) f32 cuda tensor (one fragment, all 60 group elements); complete on the caller's
        stream (the side lane is joined before the return).  join=False (Feature_extracting): the caller's stream is NOT made to
        wait for the side lane - the next fragment's first pass can then be queued under this fragment's last one; the result is
        complete once BOTH streams of `_lanes()` have run."""
        nb = 15                                                      # rotated copies per backbone pass (split further by voxel count)
        lanes = self._lanes()
        main = lanes[0][1]
        cur = torch.cuda.current_stream()
        if main is not cur:
            main.wait_stream(cur)                                    # lane 0 is a stream of our own (the caller is on the null stream)
        with torch.cuda.stream(main):
            pc_d = torch.from_numpy(np.ascontiguousarray(np.asarray(pc, dtype=np.float64))).cuda()
            k_d = torch.from_numpy(np.ascontiguousarray(np.asarray(keys, dtype=np.float64))).cuda()
            out = torch.empty((k_d.shape[0], 32, 60), dtype=torch.float32, device="cuda")
        ready = torch.cuda.Event()
        ready.record(main)
        for c, st in lanes:
            c.set_nn_grid(self.config.voxel_size)                    # the targets are one point per voxel: grid search, same winners
            if st is not main:
                st.wait_event(ready)
        try:
            for b, g0 in enumerate(range(0, 60, nb)):
                ctx, st = lanes[b % len(lanes)]
                with torch.cuda.stream(st):                          # the pass's tensors live and die on its lane's stream
                    # rotated copies (pc @ R_g^T, :143) are never materialised: rotation, voxelisation and 'dspcd0' (the
                    # down-sampled points, .float(), :92) come out of one pass over the cloud
                    res = self.fcgf.extract_rotated_batch(pc_d, [self.Rgroup[g] for g in range(g0, g0 + nb)], self.config.voxel_size, ctx=ctx)
                    for j, (sel, feat, pts) in enumerate(res):
                        ctx.group_gather(k_d, pts, feat, g0 + j, out)    # keys @ R_g^T, f64 NN, feature row -> out[:, :, g]
                    del res
            for _, st in lanes[1:]:
                if join:
                    main.wait_stream(st)
                else:                                                # still in use there when this frame's references are dropped
                    for t in (pc_d, k_d, out):
                        t.record_stream(st)
            if join and main is not cur:
                cur.wait_stream(main)                                # complete on the caller's stream
                for t in (pc_d, k_d, out):
                    t.record_stream(cur)
        except BaseException:
            for _, st in lanes:                                      # queued work still uses the tensors of this frame
                st.synchronize()
            raise
        finally:
            for c, _ in lanes:
                c.set_nn_grid(0)
        return out

    def Feature_extracting(self):
        import time
        jobs = []
        for scene, dataset in self.datasets.items():
            if scene == 'wholesetname':
                continue
            save_dir = f'{self.output_dir}/Testset/{self.dataset_name}/{scene}/FCGF_Input_Group_feature'
            make_non_exists_dir(save_dir)
            jobs += [(dataset, pc_id, f'{save_dir}/{pc_id}.npy') for pc_id in dataset.pc_ids]
        t_start = time.perf_counter()
        depth = 2                                                    # fragments read ahead / results waiting for the writer
        loaded = queue.Queue(maxsize=depth)
        towrite = queue.Queue(maxsize=depth)
        errors = []

        def loader():
            try:
                for dataset, pc_id, fn in jobs:
                    loaded.put((dataset.get_pc(pc_id), dataset.get_kps(pc_id), fn))
            except BaseException as e:                               # handed to the main thread
                errors.append(e)
            loaded.put(None)

        def writer():
            while True:
                item = towrite.get()
                if item is None:
                    return
                host, done, fn = item
                try:
                    done.synchronize()
                    np.save(fn, host.numpy())
                except BaseException as e:
                    errors.append(e)

        tl, tw = threading.Thread(target=loader, daemon=True), threading.Thread(target=writer, daemon=True)
        tl.start(); tw.start()
        copy_stream = hip.concurrent_stream(self.ctx, [st for _, st in self._lanes()])
        n = 0
        try:
            while True:
                item = loaded.get()
                if item is None or errors:
                    break
                pc, kps, fn = item
                out = self.fragment_group_features(pc, kps, join=False)
                host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
                for _, st in self._lanes():
                    copy_stream.wait_stream(st)                      # the fragment is complete when every lane has run
                with torch.cuda.stream(copy_stream):
                    host.copy_(out, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(copy_stream)
                out.record_stream(copy_stream)
                del out
                towrite.put((host, done, fn))                        # blocks while `depth` results are still waiting: bounds the pinned bytes
                n += 1
        finally:
            towrite.put(None)
            tw.join()
            if tl.is_alive():                                        # stopped early: let the loader run out of its bounded queue
                while loaded.get() is not None:
                    pass
            tl.join()
        torch.cuda.synchronize()
        if errors:
            raise errors[0]
        self.stats = {"fragments": n, "seconds": time.perf_counter() - t_start, "lanes": self.lanes}

    def batch_feature_extraction(self):
        self.Feature_extracting()
