"""Dataset-level driver: every scene pair of a test set over N GPUs (one process per GPU).

The reference's unit of work is the scene loop of tests/evaluator.py:75-101 / :146-173 (one scene after the other:
extract, match, coarse rotation, [PartII,] estimate, then the scene's pre.log) plus, for YOHO-C, a process pool over
the pairs of a scene (tests/estimator.py:255-275).  Nothing in that loop couples two pairs, so it shards with no
data-path collective:

  * scenes are dealt to the ranks whole, largest first onto the least loaded rank; a scene with more pairs than one
    rank's share (ceil(total pairs / world), e.g. 3DMatch 'kitchen' among utils/dataset.py:163-167's eight scenes) is
    cut into several parts whose pairs go round-robin over the least loaded ranks (`plan_shards`);
  * a rank keeps what its pairs need resident in HBM: the FCGF group features of the fragments it touches, their PartI
    descriptors (computed once per fragment per rank) and keypoints (`ScenePairRunner`); every pair is then one
    pipeline.run_pair call with no disk stage in between;
  * the checkpoints are read by rank 0 and broadcast once (dist.broadcast_state_dict: RCCL over xGMI);
  * each pair yields (trans (3,4), recalltime); the per-rank dictionaries are gathered on the host and rank 0 writes
    the reference's files - `{id0}-{id1}.npz` and one `pre.log` per scene, in dataset.pair_ids order - under
    Match/YOHO_{C,O}/{max_iter}iters, where utils/RR_cal.py expects them (`write_scene_results`).

Randomness.  The reference's estimators consume the global np.random stream pair after pair (YOHO-O) or restart every
pair from the same state (yohoc_mul's fork).  A sharded run cannot share one stream, so every pair draws from its own:
seed = pair_seed(base_seed, dataset.name, id0, id1).  The result of a pair therefore does not depend on the number of
ranks or on which rank ran it (tests/test_dist_cpu.py checks world 2 == world 1).
"""
import os
import threading
import zlib
import numpy as np

from . import dist as ydist
from .estimator import format_log_entry, NO_ESTIMATE  # noqa: F401  (re-exported for callers)


# ---------------------------------------------------------------------------------------------------------------
# work plan
# ---------------------------------------------------------------------------------------------------------------
FRAG_COST = 4.0       # cost of loading + describing one fragment in units of one pair (disk + H2D + PartI ~ 6 ms against ~1.5 ms per pair;
                      # tools/bench_dataset.py measures both)
MIN_PART = 8          # a scene is not cut into parts of fewer pairs than this (except its last remainder)


def part_cost(n_pairs, n_frags=None, frag_cost=FRAG_COST):
    """cost model of one (rank, scene part): its pairs + the fragments it has to load and describe - at most all of the scene's, at
    most two per pair (upper bound; contiguous pairs of a sorted gt.log share fragments)"""
    n_pairs = int(n_pairs)
    if n_pairs <= 0:
        return 0.0
    touched = 2 * n_pairs if n_frags is None else min(int(n_frags), 2 * n_pairs)
    return n_pairs + frag_cost * touched


def _fit_pairs(room, n_frags, frag_cost):
    """largest k with part_cost(k) <= room"""
    if room <= 0:
        return 0
    if n_frags is not None and room >= part_cost(-(-int(n_frags) // 2), n_frags, frag_cost):
        return int(room - frag_cost * int(n_frags))
    return int(room / (1.0 + 2.0 * frag_cost))


def _plan_wrap(order, frags, frag_cost, world, T):
    """fill rank 0, 1, ... up to cost T each, a scene that does not fit spilling its remaining pairs into the next rank
    (McNaughton's wrap-around rule with a per-part overhead); None if `world` ranks of capacity T do not hold everything"""
    plan = [[] for _ in range(world)]
    r, load = 0, 0.0
    for scene, n in order:
        f = frags.get(scene)
        start = 0
        while start < n:
            if r >= world:
                return None
            rest = n - start
            if load + part_cost(rest, f, frag_cost) <= T:
                take = rest
            else:
                take = min(_fit_pairs(T - load, f, frag_cost), rest - MIN_PART)      # leave a sensible remainder
                if take < MIN_PART:                                                  # this rank is full
                    r, load = r + 1, 0.0
                    continue
            plan[r].append((scene, list(range(start, start + take))))
            load += part_cost(take, f, frag_cost)
            start += take
    return plan


def _plan_lpt(order, frags, frag_cost, world, target):
    """scenes whole, largest first onto the least loaded rank; only a scene that costs more than the per-rank target is cut, into
    the fewest equal contiguous blocks that fit under it"""
    plan = [[] for _ in range(world)]
    load = [0.0] * world
    for scene, n in order:
        f = frags.get(scene)
        parts = 1
        while parts < min(world, max(1, n // MIN_PART)) and part_cost(-(-n // parts), f, frag_cost) > target:
            parts += 1
        bounds = [n * j // parts for j in range(parts + 1)]
        ranks = sorted(range(world), key=lambda q: (load[q], q))[:parts]
        for j, r in enumerate(ranks):
            pos = list(range(bounds[j], bounds[j + 1]))
            if pos:
                plan[r].append((scene, pos))
                load[r] += part_cost(len(pos), f, frag_cost)
    return plan


def plan_shards(scene_sizes, world, scene_frags=None, frag_cost=FRAG_COST):
    """scene_sizes: {scene key: number of pairs}; scene_frags: {scene key: number of fragments} (optional: without it a part is
    charged two fragments per pair).  Returns plan[rank] = list of (scene key, [pair positions]), the positions indexing
    dataset.pair_ids.  Deterministic; every (scene, position) appears exactly once.

    A part is a CONTIGUOUS block of a scene's pairs (neighbouring pairs of a gt.log share fragments, so a block loads fewer
    fragments than a strided subset).  Two candidate plans under the cost model of part_cost, the one with the smaller maximum
    rank load wins: (a) scenes whole, largest first onto the least loaded rank, only scenes above the per-rank target cut;
    (b) wrap-around filling at the smallest per-rank capacity that holds everything (binary search), which cuts a scene at every
    rank boundary and pays the fragments of both parts for it."""
    world = max(1, int(world))
    frags = scene_frags or {}
    order = sorted(((k, int(n)) for k, n in scene_sizes.items() if int(n) > 0), key=lambda kv: (-kv[1], str(kv[0])))
    if not order:
        return [[] for _ in range(world)]
    total = sum(part_cost(n, frags.get(k), frag_cost) for k, n in order)
    if world == 1:
        return [[(k, list(range(n))) for k, n in order]]
    best = _plan_lpt(order, frags, frag_cost, world, total / world)
    lo, hi = total / world, total
    wrap = None
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        p = _plan_wrap(order, frags, frag_cost, world, mid)
        if p is None:
            lo = mid
        else:
            wrap, hi = p, mid
    if wrap is not None and max(plan_loads(wrap, frags, frag_cost)) < max(plan_loads(best, frags, frag_cost)) - 1e-9:
        best = wrap
    return best


def plan_loads(plan, scene_frags=None, frag_cost=FRAG_COST):
    """cost of every rank's share under part_cost (tests, bench: the max / mean ratio is the planned imbalance)"""
    frags = scene_frags or {}
    return [sum(part_cost(len(pos), frags.get(s), frag_cost) for s, pos in part) for part in plan]


def pair_seed(base_seed, scene_name, id0, id1):
    """63-bit seed of one pair, the same whatever rank runs it"""
    h = zlib.crc32(f"{scene_name}|{id0}|{id1}".encode())
    x = (int(base_seed) * 0x9E3779B97F4A7C15 + h * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & (2 ** 64 - 1)
    x ^= x >> 31
    x = (x * 0xD6E8FEB86659FD93) & (2 ** 64 - 1)
    x ^= x >> 32
    return x & (2 ** 63 - 1)


def scene_items(datasets):
    """(key, dataset) of a get_dataset()-style dict, without its 'wholesetname' entry"""
    return [(k, d) for k, d in datasets.items() if k != 'wholesetname']


def run_sharded(datasets, pair_fn, rank=0, world=1, scene_fn=None, gather=None, pairs_fn=None, parts_fn=None):
    """Run pair_fn(dataset, (id0, id1)) -> picklable result for every pair of every scene, sharded by plan_shards.
    scene_fn(dataset, pairs) is called once per (rank, scene part) before its pairs (descriptor extraction).
    pairs_fn(dataset, pairs) -> list of results (same order), when given, takes a whole scene part instead of pair_fn being
    called pair by pair (the GPU worker runs several pairs concurrently).
    parts_fn([(dataset, pairs), ...]) -> [list of results per part], when given, takes ALL of the rank's scene parts at once (the GPU
    worker overlaps one part's pairs with the next part's descriptor extraction).
    Returns on EVERY rank {scene key: [result per pair, in dataset.pair_ids order]} (gathered on the host)."""
    items = scene_items(datasets)
    plan = plan_shards({k: len(d.pair_ids) for k, d in items}, world, {k: len(d.pc_ids) for k, d in items})
    by_key = dict(items)
    mine = {}
    try:
        if parts_fn is not None:
            todo = [(key, positions, by_key[key], [tuple(by_key[key].pair_ids[p]) for p in positions]) for key, positions in plan[rank]]
            for (key, positions, _, _), results in zip(todo, parts_fn([(ds, pairs) for _, _, ds, pairs in todo])):
                for p, res in zip(positions, results):
                    mine[(key, p)] = res
        for key, positions in (plan[rank] if parts_fn is None else []):
            ds = by_key[key]
            pairs = [tuple(ds.pair_ids[p]) for p in positions]
            if scene_fn is not None:
                scene_fn(ds, pairs)
            if pairs_fn is not None:
                for p, res in zip(positions, pairs_fn(ds, pairs)):
                    mine[(key, p)] = res
            else:
                for p, pair in zip(positions, pairs):
                    mine[(key, p)] = pair_fn(ds, pair)
    except Exception as e:
        # a rank that fails must still reach the gather, or the other ranks would wait for it forever: the error travels with
        # the results and is raised on EVERY rank
        import traceback
        mine = {("__error__", rank): f"rank {rank}: {type(e).__name__}: {e}\n{traceback.format_exc()}"}
    parts = (gather or ydist.gather_results)(mine)
    errors = [v for part in parts for kp, v in part.items() if kp[0] == "__error__"]
    if errors:
        raise RuntimeError("sharded run failed:\n" + "\n".join(errors))
    merged = {}
    for part in parts:
        for kp, v in part.items():
            if kp in merged:
                raise RuntimeError(f"pair {kp} was processed twice")
            merged[kp] = v
    out = {}
    for key, ds in items:
        missing = [p for p in range(len(ds.pair_ids)) if (key, p) not in merged]
        if missing:
            raise RuntimeError(f"scene {key}: {len(missing)} pairs were not processed (first: {ds.pair_ids[missing[0]]})")
        out[key] = [merged[(key, p)] for p in range(len(ds.pair_ids))]
    return out


def result_dir(cfg, dataset, yoho_sign, max_iter):
    return f'{cfg.output_cache_fn}/Testset/{dataset.name}/Match/{yoho_sign}/{max_iter}iters'


def save_pair_npz(save_dir, id0, id1, res):
    """{id0}-{id1}.npz as tests/estimator.py:131-137,337-339 leaves it"""
    np.savez(f'{save_dir}/{id0}-{id1}.npz', trans=res["trans"], recalltime=res["recalltime"])


def write_scene_results(cfg, dataset, results, yoho_sign, max_iter, npz=True):
    """the files tests/estimator.py leaves for one scene: {id0}-{id1}.npz per pair + pre.log in pair order (:12-24).
    npz=False: the per-pair archives were already written by the ranks that ran the pairs (ScenePairRunner.finish_writes)."""
    save_dir = result_dir(cfg, dataset, yoho_sign, max_iter)
    os.makedirs(save_dir, exist_ok=True)
    text = []
    for (id0, id1), res in zip(dataset.pair_ids, results):
        if npz:
            save_pair_npz(save_dir, id0, id1, res)
        text.append(format_log_entry(id0, id1, len(dataset.pc_ids), res["trans"]))
    with open(f'{save_dir}/pre.log', 'w') as f:
        f.write("".join(text))
    return save_dir


# ---------------------------------------------------------------------------------------------------------------
# the GPU worker of one rank
# ---------------------------------------------------------------------------------------------------------------
class _Part:
    """the resident fragments of one scene part: fid -> dict(feat, keys, eqv, inv_np), how many pairs still need each, the
    event behind which each is described (frag_ev: set, under `cond`, when its PartI pass has been queued) and the order in which
    the fragments are described (rank_of), from which the pair workers take the pairs in the order they become runnable"""

    def __init__(self, scene, pairs=()):
        self.scene = scene
        self.frag = {}
        self.frag_ev = {}
        self.uses = {}
        self.cond = threading.Condition()
        self.failed = False
        for p in pairs:
            for i in p:
                self.uses[i] = self.uses.get(i, 0) + 1
        self.need = sorted(self.uses, key=lambda v: int(v))
        self.rank_of = {fid: k for k, fid in enumerate(self.need)}

    def wait_for(self, pair):
        """block until both fragments of the pair are described (their passes queued) -> the events to wait for on the device"""
        with self.cond:
            while not self.failed and not all(f in self.frag_ev for f in pair):
                self.cond.wait(0.5)
            if self.failed:
                raise RuntimeError(f"scene part {self.scene}: loading / describing its fragments failed")
            return [self.frag_ev[f] for f in pair]

    def fail(self):
        with self.cond:
            self.failed = True
            self.cond.notify_all()


class ScenePairRunner:
    """HBM-resident execution of the pairs one rank owns.  estimator 'yohoo' (needs the PartII weights) or 'yohoc'.

    A fragment stays resident (FCGF group feature, PartI descriptor, keypoints: 77 MB at 5000 keypoints) exactly as long as a
    pair of its scene part still needs it: the part counts the uses, run_pair releases a fragment after its last pair, and
    run_parts keeps at most two parts in flight (one whose pairs run, one being loaded and described), so a rank's footprint is
    bounded by two scene parts whatever the number of parts it walks.
    `stats` accumulates where the time goes (seconds; device work is timed with a synchronise only when timing=True)."""

    def __init__(self, cfg, ctx, estimator="yohoo", max_iter=1000, base_seed=0, timing=False, write_npz=False, hypotheses="selected",
                 pair_workers=2, partII_sd=None, fused=True, overlap=True, backbone=None):
        import torch
        from . import pipeline
        self.torch, self.pipeline = torch, pipeline
        self.cfg, self.ctx = cfg, ctx
        self.estimator, self.max_iter, self.base_seed = estimator, int(max_iter), int(base_seed)
        self.inlier_dist = cfg.ransac_o_inlinerdist if estimator == "yohoo" else cfg.ransac_c_inlinerdist
        # YOHO-O: PartII only for the <= max_iter matches the vote reads (pipeline.run_pair, hypotheses="selected"): the driver's
        # products are trans / recalltime per pair, identical either way; "all" computes Trans_pre for every match as the
        # reference's stage does
        self.hypotheses = hypotheses
        # fused: a pair is ONE library call (yoho_register_pair; the GIL is released for its whole duration, so the pair workers
        # really overlap) instead of pipeline.run_pair's ~25 calls from Python - the same entries in the same order with the same
        # vote order / sampling stream, hence the same bits (tests/test_gpu_dropin.py); False keeps the Python composition
        self.fused = bool(fused)
        # overlap: run_parts lets pairs run while further fragments are loaded and described; False = part by part, setup then pairs
        self.overlap = bool(overlap)
        # pair_workers > 1: run_pairs runs that many pairs at a time, each on its own HIP stream with its own library context
        # (a context is single-stream by contract; PartI is not needed there, PartII's weights come from partII_sd).  A pair's
        # kernels at <= 1000 voted matches are a chain of ~25 short launches that fill a fraction of the chip and end in two
        # host read-backs: two chains side by side hide both.  Results do not depend on the interleaving (per-pair seeds, no
        # shared state between pairs).
        # own_contexts: the pairs can run in library contexts of their own (YOHO-C needs no weights there, YOHO-O the PartII state
        # dict); only then may pairs run while the caller's context describes further fragments (run_parts)
        self.own_contexts = estimator == "yohoc" or partII_sd is not None
        self.pair_workers = max(1, int(pair_workers)) if self.own_contexts else 1
        self._partII_sd = partII_sd
        self._workers = None            # [(context, torch stream)]
        self._lock = threading.Lock()
        self.part = _Part(None)         # the scene part set up last (run_parts keeps two alive: one being set up, one being run)
        self._made_dirs = set()
        self._copy_stream = None
        # backbone: an object with fragment_group_features(pc (N,3), keys (K,3)) -> (K,32,60) f32 on the device, complete on the current
        # stream (YOHO_testset.testset_create): the FCGF group features then come from the fragments' point clouds - rotated, voxelised,
        # through the sparse backbone, NN-gathered at the keypoints, all on this GPU - instead of from the FCGF_Input_Group_feature cache
        # files of a separate YOHO_testset.py run.  Nothing between the raw cloud and the registration result touches the disk.
        self.backbone = backbone
        self._pin_pool = {}
        self._pin_lock = threading.Lock()
        # write_npz: this rank writes the {id0}-{id1}.npz of the pairs it ran (finish_writes, behind its last pair: 0.15 ms of
        # Python per archive, spread over the ranks instead of serial on rank 0; a worker thread writing them WHILE the pairs run
        # was measured and dropped - it saved 0.07 s per 495 pairs and cost every pair 0.1 ms of GIL contention)
        self.yoho_sign = 'YOHO_O' if estimator == "yohoo" else 'YOHO_C'
        self._write_npz = bool(write_npz)
        self._write_jobs = []
        self.timing = bool(timing)
        self.stats = {"fragments": 0, "pairs": 0, "load_s": 0.0, "load_wait_s": 0.0, "h2d_describe_s": 0.0, "setup_s": 0.0, "pairs_s": 0.0,
                      "bytes_read": 0, "peak_resident_fragments": 0, "backbone_s": 0.0}

    def _feature_dir(self, dataset):
        from .utils import dataset_feature_name
        return f'{self.cfg.output_cache_fn}/Testset/{dataset_feature_name(dataset.name)}/FCGF_Input_Group_feature'

    def _pinned(self, shape, dtype):
        """a page-locked staging tensor from a small pool (allocating pinned memory per fragment costs more than reading the file)"""
        torch = self.torch
        key = (tuple(shape), dtype)
        with self._pin_lock:
            pool = self._pin_pool.setdefault(key, [])
            for i, (buf, ev) in enumerate(pool):
                if ev is None or ev.query():                 # its last upload has completed
                    pool.pop(i)
                    return buf
        return torch.empty(shape, dtype=dtype).pin_memory()

    def _recycle(self, buf, event):
        with self._pin_lock:
            self._pin_pool.setdefault((tuple(buf.shape), buf.dtype), []).append((buf, event))

    def _load_fragment(self, dataset, fdir, fid):
        """disk -> page-locked host tensors: the .npy payload is read straight into a pinned buffer (header parsed with numpy's own
        reader, then one readinto: no intermediate array, no page-fault-driven mmap copy)"""
        import time
        torch = self.torch
        t0 = time.perf_counter()
        with open(f'{fdir}/{fid}.npy', 'rb') as f:
            major, minor = np.lib.format.read_magic(f)
            shape, fortran, dt = (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(f)
            if fortran or dt != np.dtype('<f4'):
                x = torch.from_numpy(np.ascontiguousarray(np.load(f'{fdir}/{fid}.npy'), dtype=np.float32)).pin_memory()
            else:
                x = self._pinned(shape, torch.float32)
                view = memoryview(x.numpy()).cast('B')
                got = 0
                while got < len(view):
                    n = f.readinto(view[got:])
                    if not n:
                        raise IOError(f'{fdir}/{fid}.npy: file shorter than its header says')
                    got += n
        keys = torch.from_numpy(np.ascontiguousarray(dataset.get_kps(fid), dtype=np.float64)).pin_memory()
        return fid, x, keys, time.perf_counter() - t0

    def setup_scene(self, dataset, pairs, part=None):
        """load + describe every fragment the pairs touch (tests/extractor.py:37-62 without the .npy round trip): a loader
        thread reads the cache files into pinned memory while the device describes the fragments already there; up to 16384
        keypoints (3 fragments of 5000) go through one PartI pass"""
        import queue
        import threading
        import time
        torch = self.torch
        t_setup = time.perf_counter()
        if part is None:
            part = _Part(dataset.name, pairs)
        need = part.need
        if self.backbone is not None:
            return self._setup_scene_from_clouds(dataset, part, t_setup)
        fdir = self._feature_dir(dataset)
        q = queue.Queue(maxsize=6)
        NLOAD = 3                                       # loader threads (file reads release the GIL); results are consumed in order

        stop = threading.Event()                        # set when the consumer gives up (its exception must not leave the loader blocked)

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    pass
            return False

        def loader():
            from concurrent.futures import ThreadPoolExecutor
            try:
                # loader threads start on device 0: pin_memory() there would create a primary context on GPU 0 in every rank's process
                with ThreadPoolExecutor(NLOAD, initializer=torch.cuda.set_device, initargs=(self.ctx.device,)) as ex:
                    pending = []
                    for fid in need:
                        if stop.is_set():
                            break
                        pending.append(ex.submit(self._load_fragment, dataset, fdir, fid))
                        if len(pending) >= NLOAD + 1 and not put(pending.pop(0).result()):
                            break
                    for fut in pending:
                        if not put(fut.result()):
                            break
            except BaseException as e:          # surfaced in the consumer
                put(e)
            put(None)
        th = threading.Thread(target=loader, daemon=True)
        th.start()

        # two-stage device side: the H2D copies of group N + 1 run on a copy stream while PartI describes group N
        if self._copy_stream is None:
            from . import hip
            # (streams are dealt round-robin onto four hardware queues: one picked blindly may run strictly behind the caller's)
            self._copy_stream = hip.concurrent_stream(self.ctx, [torch.cuda.current_stream()])
        copy, main = self._copy_stream, torch.cuda.current_stream()

        def upload(group):
            with torch.cuda.stream(copy):
                dev = [(fid, x.cuda(non_blocking=True), keys.cuda(non_blocking=True)) for fid, x, keys in group]
                up = torch.cuda.Event()
                up.record(copy)
            for _, x, _ in group:
                if x.is_pinned():
                    self._recycle(x, up)
            return dev, up

        def describe(dev, up):
            t0 = time.perf_counter()
            main.wait_event(up)
            for _, x, keys in dev:                  # allocated on the copy stream, used (and later freed) on this one
                x.record_stream(main)
                keys.record_stream(main)
            xs = torch.cat([g[1] for g in dev]) if len(dev) > 1 else dev[0][1]
            out = self.ctx.partI_forward(xs.contiguous(), want_inv=False, want_inv_np=True)
            o = 0
            for fid, x, keys in dev:
                n = x.shape[0]
                # own storage per fragment, so that releasing one fragment frees its memory (slices would pin the whole pass)
                part.frag[fid] = dict(feat=x, keys=keys, eqv=out["eqv"][o:o + n].clone(), inv_np=out["inv_np"][o:o + n].clone())
                o += n
            self.stats["peak_resident_fragments"] = max(self.stats["peak_resident_fragments"], len(part.frag))
            ev = torch.cuda.Event()
            ev.record(main)
            with part.cond:                         # pairs of these fragments may start (run_parts: the workers are already waiting)
                for fid, _, _ in dev:
                    part.frag_ev[fid] = ev
                part.cond.notify_all()
            self.stats["h2d_describe_s"] += time.perf_counter() - t0
            self.stats["fragments"] += len(dev)

        pending = [None]

        def flush(group):
            u = upload(group)
            if pending[0] is not None:
                describe(*pending[0])
            pending[0] = u

        group, rows = [], 0
        try:
            while True:
                t0 = time.perf_counter()
                item = q.get()
                self.stats["load_wait_s"] += time.perf_counter() - t0
                if isinstance(item, BaseException):
                    raise item
                if item is not None:
                    self.stats["load_s"] += item[3]
                    self.stats["bytes_read"] += item[1].numel() * 4 + item[2].numel() * 8
                    item = item[:3]
                # a pass is flushed when full, when the input ends, or when the loader has nothing ready (do not idle the device)
                if group and (item is None or rows + item[1].shape[0] > 16384):
                    flush(group)
                    group, rows = [], 0
                if item is None:
                    break
                group.append(item)
                rows += item[1].shape[0]
                if q.empty() and group and rows >= 4096:
                    flush(group)
                    group, rows = [], 0
            if pending[0] is not None:
                describe(*pending[0])
        except BaseException:
            stop.set()                              # unblock the loader, then let the error travel
            th.join(timeout=5.0)
            part.fail()
            raise
        if self.timing:
            t0 = time.perf_counter()
            main.synchronize()                      # once per scene part: the device side of the setup is inside setup_s (this
            self.stats["h2d_describe_s"] += time.perf_counter() - t0        # stream only: the pairs of the part before may be running)
        th.join()
        self.stats["peak_resident_fragments"] = max(self.stats["peak_resident_fragments"], len(part.frag))
        self.stats["setup_s"] += time.perf_counter() - t_setup
        self.part = part
        return part

    def _describe_group(self, part, dev):
        """PartI over the fragments of `dev` = [(fid, group feature (K,32,60) cuda, keys (K,3) f64 cuda)], all valid on the current
        stream; the fragments become resident in `part` and their pairs may start"""
        torch = self.torch
        main = torch.cuda.current_stream()
        xs = torch.cat([g[1] for g in dev]) if len(dev) > 1 else dev[0][1]
        out = self.ctx.partI_forward(xs.contiguous(), want_inv=False, want_inv_np=True)
        o = 0
        for fid, x, keys in dev:
            n = x.shape[0]
            part.frag[fid] = dict(feat=x, keys=keys, eqv=out["eqv"][o:o + n].clone(), inv_np=out["inv_np"][o:o + n].clone())
            o += n
        ev = torch.cuda.Event()
        ev.record(main)
        with part.cond:
            for fid, _, _ in dev:
                part.frag_ev[fid] = ev
            part.cond.notify_all()
        self.stats["fragments"] += len(dev)
        self.stats["peak_resident_fragments"] = max(self.stats["peak_resident_fragments"], len(part.frag))

    def _setup_scene_from_clouds(self, dataset, part, t_setup):
        """setup_scene with the group features computed here (self.backbone) instead of read from the cache: per fragment the
        backbone's sixty passes (two lanes), then PartI; the pairs of fragments already described run meanwhile on the worker streams"""
        import time
        torch = self.torch
        main = torch.cuda.current_stream()
        try:
            for fid in part.need:
                t0 = time.perf_counter()
                keys_h = np.ascontiguousarray(dataset.get_kps(fid), dtype=np.float64)
                x = self.backbone.fragment_group_features(dataset.get_pc(fid), keys_h)
                keys = torch.from_numpy(keys_h).cuda()
                self.stats["backbone_s"] += time.perf_counter() - t0
                t0 = time.perf_counter()
                self._describe_group(part, [(fid, x, keys)])
                self.stats["h2d_describe_s"] += time.perf_counter() - t0
        except BaseException:
            part.fail()
            raise
        if self.timing:
            t0 = time.perf_counter()
            main.synchronize()
            self.stats["h2d_describe_s"] += time.perf_counter() - t0
        self.stats["setup_s"] += time.perf_counter() - t_setup
        self.part = part
        return part

    def finish_writes(self):
        """write the archives of the pairs this rank ran"""
        for save_dir, id0, id1, res in self._write_jobs:
            if save_dir not in self._made_dirs:
                os.makedirs(save_dir, exist_ok=True)
                self._made_dirs.add(save_dir)
            save_pair_npz(save_dir, id0, id1, res)
        self._write_jobs = []

    # the part set up last, under the names the single-part interface (setup_scene -> run_pair / run_pairs) has always used
    @property
    def frag(self):
        return self.part.frag

    @property
    def uses(self):
        return self.part.uses

    @staticmethod
    def _release(part, fid):
        n = part.uses.get(fid, 0) - 1
        if n <= 0:
            part.uses.pop(fid, None)
            part.frag.pop(fid, None)
        else:
            part.uses[fid] = n

    def run_pair(self, dataset, pair, ctx=None, part=None):
        import time
        t0 = time.perf_counter()
        id0, id1 = pair
        part = part if part is not None else self.part
        a, b = part.frag[id0], part.frag[id1]
        seed = pair_seed(self.base_seed, dataset.name, id0, id1)
        c = ctx if ctx is not None else self.ctx
        out = f = None
        if self.fused and (self.estimator == "yohoc" or c.supports_matched()):
            f = c.register_pair(a["feat"], b["feat"], a["eqv"], b["eqv"], a["inv_np"], b["inv_np"], a["keys"], b["keys"], estimator=self.estimator,
                                max_iter=self.max_iter, inlier_dist=self.inlier_dist, seed=seed, selected=(self.hypotheses == "selected"))
            if not f["range_flag"]:
                out = {"trans": f["trans"], "recalltime": f["best_h"], "matches": f["matches"], "inliers": f["best_count"]}
        if out is None:
            compose = lambda: self.pipeline.run_pair(
                c, a["feat"], b["feat"], a["keys"], b["keys"], inlier_dist=self.inlier_dist, max_iter=self.max_iter,
                order_rng=np.random.RandomState(seed & 0xFFFFFFFF), eqv=({"eqv": a["eqv"], "inv_np": a["inv_np"]}, {"eqv": b["eqv"], "inv_np": b["inv_np"]}),
                estimator=self.estimator, seed=seed, hypotheses=self.hypotheses)
            if f is not None:
                # the one-call pair reported a value outside the fp16 range of PartII (the call consumed the device flag): the pair is
                # composed ONCE more with this worker's PartII in bf16x3 - not a second fp16x2 attempt that would have to overflow
                # again to be noticed; after hip.Context.range_sticky_after such pairs the worker stays in bf16x3 (range_report)
                r = c._repeat_wider("partII", compose)
            else:
                r = compose()
            out = {"trans": np.asarray(r.trans, dtype=np.float64), "recalltime": int(r.best_h), "matches": int(r.match.shape[0]),
                   "inliers": int(r.best_count)}
        with self._lock:
            self._release(part, id0)
            self._release(part, id1)
            if self._write_npz:
                self._write_jobs.append((result_dir(self.cfg, dataset, self.yoho_sign, self.max_iter), id0, id1, out))
            self.stats["pairs"] += 1
            self.stats["pairs_s"] += time.perf_counter() - t0
        return out

    def _make_workers(self):
        from . import hip
        torch = self.torch
        ws = []
        for _ in range(self.pair_workers):
            c = hip.Context(self.ctx.device, getattr(self.ctx.tables, "dir", None))
            c.set_partII_mode(self.ctx.partII_mode)
            if self.estimator == "yohoo":
                c.load_partII(self._partII_sd)
            # a worker's stream must overlap the caller's (fragments are described there meanwhile) and the other workers': measured
            ws.append((c, hip.concurrent_stream(self.ctx, [torch.cuda.current_stream()] + [st for _, st in ws])))
        return ws

    def run_pairs(self, dataset, pairs, part=None):
        """the pairs of a scene part, pair_workers at a time -> results in the order of `pairs`"""
        import time
        pairs = list(pairs)
        part = part if part is not None else self.part
        torch = self.torch
        # in the order the pairs become runnable: by the later of their two fragments in the order of description
        order = sorted(range(len(pairs)), key=lambda i: max(part.rank_of.get(pairs[i][0], 0), part.rank_of.get(pairs[i][1], 0)))
        if not self.own_contexts:                        # on the caller's context and stream: nothing else may use them meanwhile
            res = [None] * len(pairs)
            for i in order:
                for ev in part.wait_for(pairs[i]):
                    torch.cuda.current_stream().wait_event(ev)
                res[i] = self.run_pair(dataset, pairs[i], part=part)
            return res
        if self._workers is None:
            self._workers = self._make_workers()
        out = [None] * len(pairs)
        nxt = [0]
        errors = []
        t_wall = time.perf_counter()

        def work(wi):
            c, st = self._workers[wi]
            try:
                torch.cuda.set_device(self.ctx.device)
                with torch.cuda.stream(st):
                    while not errors:
                        with self._lock:
                            k = nxt[0]
                            nxt[0] += 1
                        if k >= len(order):
                            break
                        i = order[k]
                        for ev in part.wait_for(pairs[i]):      # blocks while the fragments are still being loaded / described
                            st.wait_event(ev)
                        out[i] = self.run_pair(dataset, pairs[i], ctx=c, part=part)
                    st.synchronize()
            except BaseException as e:
                errors.append(e)
        ths = [threading.Thread(target=work, args=(wi,), daemon=True) for wi in range(min(len(self._workers), max(1, len(pairs))))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            raise errors[0]
        self.stats["pairs_wall_s"] = self.stats.get("pairs_wall_s", 0.0) + time.perf_counter() - t_wall
        return out


    def run_parts(self, parts):
        """[(dataset, pairs), ...] -> [results per part].  The pair workers of a part start BEFORE its fragments are loaded: a pair
        runs (on a worker's stream) as soon as both its fragments are described, while the caller's thread keeps loading and
        describing the rest of the part and then the next part - the descriptor pass fills the matrix pipes, the pair chains are
        short latency-bound launches that slot in between its workgroups (the overlap bench.py's pair streamer lives on).  At most
        two parts are in flight, so two parts' fragments bound the footprint."""
        import time
        out = [None] * len(parts)
        t_all = time.perf_counter()
        if not self.own_contexts or not self.overlap:   # the pairs need the caller's context (or the caller asked): one thing at a time
            for i, (ds, pairs) in enumerate(parts):
                self.setup_scene(ds, pairs)
                out[i] = self.run_pairs(ds, pairs)
            self.stats["parts_wall_s"] = self.stats.get("parts_wall_s", 0.0) + time.perf_counter() - t_all
            return out

        def start(i, part, ds, pairs, after):
            box = {}

            def body():
                try:
                    if after is not None:           # the pair workers (contexts, streams) serve one part at a time: the part
                        after.join()                # before must have run all its pairs; this part's SETUP overlaps them meanwhile
                    self.torch.cuda.set_device(self.ctx.device)
                    box["res"] = self.run_pairs(ds, pairs, part=part)
                except BaseException as e:
                    box["err"] = e
            th = threading.Thread(target=body, daemon=True)
            th.start()
            return i, th, box, part

        def finish(p):
            i, th, box, _ = p
            th.join()
            if "err" in box:
                raise box["err"]
            out[i] = box["res"]

        inflight = []
        t_last_setup = t_all
        try:
            for i, (ds, pairs) in enumerate(parts):
                while len(inflight) >= 2:
                    finish(inflight.pop(0))
                part = _Part(ds.name, pairs)
                inflight.append(start(i, part, ds, pairs, inflight[-1][1] if inflight else None))
                self.setup_scene(ds, pairs, part=part)
                t_last_setup = time.perf_counter()
            while inflight:
                finish(inflight.pop(0))
        finally:
            for p in inflight:                      # an error above: stop and collect the workers before the exception travels
                p[3].fail()
                p[1].join()
        t_end = time.perf_counter()
        self.stats["parts_wall_s"] = self.stats.get("parts_wall_s", 0.0) + t_end - t_all
        self.stats["pairs_tail_s"] = self.stats.get("pairs_tail_s", 0.0) + t_end - t_last_setup      # pairs not hidden behind a setup
        return out


def load_and_broadcast_weights(cfg, ctx, need_partII):
    """rank 0 reads {model_fn}/{PartI_train,PartII_train}/model_best.pth (tests/extractor.py:22-31,113-121); one
    broadcast per network puts the same tensors on every rank"""
    from . import weights as W
    import torch.distributed as tdist
    rank = tdist.get_rank() if (tdist.is_available() and tdist.is_initialized()) else 0
    sd1 = sd2 = None
    if rank == 0:
        sd1 = W.to_numpy_state_dict(W.load_checkpoint(f'{cfg.model_fn}/PartI_train/model_best.pth')[0])
        if need_partII:
            sd2 = W.to_numpy_state_dict(W.load_checkpoint(f'{cfg.model_fn}/PartII_train/model_best.pth')[0])
            sd2 = {k: v for k, v in sd2.items() if k in {n for n, _ in W.PARTII_SPEC}}
    sd1 = ydist.broadcast_state_dict(sd1, W.PARTI_SPEC)
    ctx.load_partI(sd1)
    if need_partII:
        sd2 = ydist.broadcast_state_dict(sd2, W.PARTII_SPEC)
        ctx.load_partII(sd2)
    return sd1, sd2


def eval_sharded(cfg, max_iter=1000, estimator="yohoo", datasets=None, base_seed=0, results_log=None, ctx=None, state_dicts=None,
                 stats_out=None, hypotheses="selected", weights_loaded=False, pair_workers=2, fused=True, overlap=True, fcgf_model=None,
                 voxel_size=0.025):
    """The sharded counterpart of Evaluator_PartI/II.eval (tests/evaluator.py:75-101,146-173): run every pair of the
    test set over the initialised process group (one rank per GPU), write npz / pre.log on rank 0 and return the
    Registration Recall there (None on the other ranks).  FCGF group features and keypoints are read from the
    reference's cache layout; descriptors, matches and hypotheses never touch the disk.  state_dicts: None = read and broadcast the
    checkpoints of cfg.model_fn, (PartI, PartII) = use these (weights_loaded=True: ctx already holds them).  pair_workers: pairs run
    concurrently per rank (ScenePairRunner.run_pairs).  stats_out: a dict that receives the
    rank's ScenePairRunner.stats and the gathered per-pair results (tools/bench_dataset.py, tests).
    fcgf_model (an FCGF checkpoint path or dict, as YOHO_testset.py's --model): the group features are not read from the cache but
    computed from the fragments' point clouds (dataset.get_pc) on this GPU - YOHO_testset.py's stage folded into the evaluation, with
    voxel_size as its --voxel_size; every rank runs the backbone for the fragments of the parts it owns."""
    import torch
    from . import hip, RR_cal
    from .dataset import get_dataset
    if datasets is None:
        datasets = get_dataset(cfg, False)
    rank, world, local = ydist.init_from_env()
    if ctx is None:
        ctx = hip.get_context(so3_dir=getattr(cfg, "SO3_related_files", None))
    if state_dicts is None:
        state_dicts = load_and_broadcast_weights(cfg, ctx, need_partII=(estimator == "yohoo"))
    elif not weights_loaded:                     # given by the caller (bench / tests): (PartI, PartII or None)
        ctx.load_partI(state_dicts[0])
        if estimator == "yohoo":
            ctx.load_partII(state_dicts[1])
    # every rank writes the archives of the pairs it ran (one node: the cache directory is shared), rank 0 the pre.log files
    backbone = None
    if fcgf_model is not None:
        import types
        from .YOHO_testset import testset_create
        backbone = testset_create(types.SimpleNamespace(model=fcgf_model, voxel_size=voxel_size, dataset=datasets.get("wholesetname", "testset"), datasets=datasets,
                                                        output_dir=cfg.output_cache_fn, origin_dir=getattr(cfg, "origin_data_dir", ".")), ctx=ctx)
    runner = ScenePairRunner(cfg, ctx, estimator=estimator, max_iter=max_iter, base_seed=base_seed, timing=stats_out is not None, write_npz=True,
                             hypotheses=hypotheses, pair_workers=pair_workers, partII_sd=(state_dicts[1] if estimator == "yohoo" else None),
                             fused=fused, overlap=overlap, backbone=backbone)
    cached = getattr(ctx, "_pair_workers_cache", None)           # worker contexts (PartII weight packing: 0.25 s each) live with ctx
    if cached is not None and cached[0] is state_dicts[1] and len(cached[1]) == runner.pair_workers:
        runner._workers = cached[1]
    guard_ctxs = lambda: [ctx] + [w for w, _ in (runner._workers or [])]
    repeats_before = sum(c.range_fallbacks for c in guard_ctxs())
    try:
        results = run_sharded(datasets, runner.run_pair, rank=rank, world=world, parts_fn=runner.run_parts)
        if runner._workers is not None:
            ctx._pair_workers_cache = (state_dicts[1], runner._workers)
        # fp16 range guard: how many passes / pairs of this run were repeated in bf16x3, and whether a network of this checkpoint now
        # stays there (hip.Context._repeat_wider) - a checkpoint that trips the guard costs ~3x and must not hide behind warnings
        reports = [c.range_report() for c in guard_ctxs()]
        runner.stats["range_guard"] = {
            "repeats_this_run": sum(c.range_fallbacks for c in guard_ctxs()) - repeats_before,
            "partI_repeats_since_checkpoint": reports[0]["partI_repeats"], "partI_stays_bf16x3": reports[0]["partI_stays_bf16x3"],
            "partII_repeats_since_checkpoint": sum(r["partII_repeats"] for r in reports),
            "partII_workers_staying_bf16x3": sum(1 for r in reports if r["partII_stays_bf16x3"])}
    finally:
        write_error = None
        try:
            runner.finish_writes()
        except Exception as e:               # reported below, symmetrically
            write_error = f"{type(e).__name__}: {e}"
    if ydist.max_over_ranks(0.0 if write_error is None else 1.0) > 0.0:
        raise RuntimeError("writing the per-pair archives failed on at least one rank" + (f": {write_error}" if write_error else ""))
    torch.cuda.synchronize()
    if stats_out is not None:
        stats_out.update(runner.stats)
        stats_out["results"] = results
    if rank != 0:
        ydist.barrier()
        return None
    try:
        sign = 'YOHO_O' if estimator == "yohoo" else 'YOHO_C'
        for key, ds in scene_items(datasets):
            write_scene_results(cfg, ds, results[key], sign, max_iter, npz=False)
        rr, flags, errors = RR_cal.benchmark(cfg, datasets, max_iter, yoho_sign=sign)
        if results_log:
            if os.path.dirname(results_log):
                os.makedirs(os.path.dirname(results_log), exist_ok=True)
            with open(results_log, 'a') as f:
                f.write(f"{datasets['wholesetname']}-{estimator}-{max_iter}iterations-{world}ranks\nMean_Registration_Recall {rr}\n\n")
    finally:
        ydist.barrier()                          # the other ranks wait here whatever happened to rank 0's files
    return rr
