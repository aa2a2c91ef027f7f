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
import zlib
import numpy as np

from . import dist as ydist
from .estimator import format_log_entry, NO_ESTIMATE  # noqa: F401  (re-exported for callers)


# ---------------------------------------------------------------------------------------------------------------
# work plan
# ---------------------------------------------------------------------------------------------------------------
def plan_shards(scene_sizes, world):
    """scene_sizes: {scene key: number of pairs}.  Returns plan[rank] = list of (scene key, [pair positions]), the
    positions indexing dataset.pair_ids.  Deterministic; every (scene, position) appears exactly once."""
    world = max(1, int(world))
    total = sum(int(n) for n in scene_sizes.values())
    share = max(1, -(-total // world))
    load = [0] * world
    plan = [[] for _ in range(world)]
    for scene, n in sorted(scene_sizes.items(), key=lambda kv: (-int(kv[1]), str(kv[0]))):
        n = int(n)
        if n == 0:
            continue
        parts = min(world, -(-n // share))
        ranks = sorted(range(world), key=lambda r: (load[r], r))[:parts]
        for j, r in enumerate(ranks):
            pos = list(range(j, n, parts))
            plan[r].append((scene, pos))
            load[r] += len(pos)
    return plan


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


def run_sharded(datasets, pair_fn, rank=0, world=1, scene_fn=None, gather=None):
    """Run pair_fn(dataset, (id0, id1)) -> picklable result for every pair of every scene, sharded by plan_shards.
    scene_fn(dataset, pairs) is called once per (rank, scene part) before its pairs (descriptor extraction).
    Returns on EVERY rank {scene key: [result per pair, in dataset.pair_ids order]} (gathered on the host)."""
    items = scene_items(datasets)
    plan = plan_shards({k: len(d.pair_ids) for k, d in items}, world)
    by_key = dict(items)
    mine = {}
    for key, positions in plan[rank]:
        ds = by_key[key]
        pairs = [tuple(ds.pair_ids[p]) for p in positions]
        if scene_fn is not None:
            scene_fn(ds, pairs)
        for p, pair in zip(positions, pairs):
            mine[(key, p)] = pair_fn(ds, pair)
    parts = (gather or ydist.gather_results)(mine)
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


def write_scene_results(cfg, dataset, results, yoho_sign, max_iter):
    """the files tests/estimator.py leaves for one scene: {id0}-{id1}.npz per pair + pre.log in pair order (:12-24)"""
    save_dir = f'{cfg.output_cache_fn}/Testset/{dataset.name}/Match/{yoho_sign}/{max_iter}iters'
    os.makedirs(save_dir, exist_ok=True)
    text = []
    for (id0, id1), res in zip(dataset.pair_ids, results):
        np.savez(f'{save_dir}/{id0}-{id1}.npz', trans=res["trans"], recalltime=res["recalltime"])
        text.append(format_log_entry(id0, id1, len(dataset.pc_ids), res["trans"]))
    with open(f'{save_dir}/pre.log', 'w') as f:
        f.write("".join(text))
    return save_dir


# ---------------------------------------------------------------------------------------------------------------
# the GPU worker of one rank
# ---------------------------------------------------------------------------------------------------------------
class ScenePairRunner:
    """HBM-resident execution of the pairs one rank owns.  estimator 'yohoo' (needs the PartII weights) or 'yohoc'."""

    def __init__(self, cfg, ctx, estimator="yohoo", max_iter=1000, base_seed=0):
        import torch
        from . import pipeline
        self.torch, self.pipeline = torch, pipeline
        self.cfg, self.ctx = cfg, ctx
        self.estimator, self.max_iter, self.base_seed = estimator, int(max_iter), int(base_seed)
        self.inlier_dist = cfg.ransac_o_inlinerdist if estimator == "yohoo" else cfg.ransac_c_inlinerdist
        self.scene = None
        self.frag = {}

    def _feature_dir(self, dataset):
        from .utils import dataset_feature_name
        return f'{self.cfg.output_cache_fn}/Testset/{dataset_feature_name(dataset.name)}/FCGF_Input_Group_feature'

    def setup_scene(self, dataset, pairs):
        """load + describe every fragment the pairs touch (tests/extractor.py:37-62 without the .npy round trip);
        up to 3 fragments of 5000 keypoints go through one PartI pass"""
        torch = self.torch
        if self.scene != dataset.name:
            self.scene, self.frag = dataset.name, {}
        need = sorted({i for p in pairs for i in p if i not in self.frag}, key=lambda v: int(v))
        fdir = self._feature_dir(dataset)
        loaded = []
        for fid in need:
            x = torch.from_numpy(np.ascontiguousarray(np.load(f'{fdir}/{fid}.npy'), dtype=np.float32)).cuda()
            keys = torch.from_numpy(np.ascontiguousarray(dataset.get_kps(fid), dtype=np.float64)).cuda()
            loaded.append((fid, x, keys))
        group, rows = [], 0
        for item in loaded + [None]:
            if item is None or (group and rows + item[1].shape[0] > 16384):
                if group:
                    xs = torch.cat([g[1] for g in group]) if len(group) > 1 else group[0][1]
                    out = self.ctx.partI_forward(xs.contiguous(), want_inv=False, want_inv_np=True)
                    o = 0
                    for fid, x, keys in group:
                        n = x.shape[0]
                        self.frag[fid] = dict(feat=x, keys=keys, eqv=out["eqv"][o:o + n], inv_np=out["inv_np"][o:o + n])
                        o += n
                group, rows = [], 0
            if item is not None:
                group.append(item)
                rows += item[1].shape[0]

    def run_pair(self, dataset, pair):
        id0, id1 = pair
        a, b = self.frag[id0], self.frag[id1]
        seed = pair_seed(self.base_seed, dataset.name, id0, id1)
        r = self.pipeline.run_pair(self.ctx, a["feat"], b["feat"], a["keys"], b["keys"], inlier_dist=self.inlier_dist,
                                   max_iter=self.max_iter, order_rng=np.random.RandomState(seed & 0xFFFFFFFF),
                                   eqv=({"eqv": a["eqv"], "inv_np": a["inv_np"]}, {"eqv": b["eqv"], "inv_np": b["inv_np"]}),
                                   estimator=self.estimator, seed=seed)
        trans = np.asarray(r.trans, dtype=np.float64)
        return {"trans": trans, "recalltime": int(r.best_h), "matches": int(r.match.shape[0]), "inliers": int(r.best_count)}


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
    ctx.load_partI(ydist.broadcast_state_dict(sd1, W.PARTI_SPEC))
    if need_partII:
        ctx.load_partII(ydist.broadcast_state_dict(sd2, W.PARTII_SPEC))


def eval_sharded(cfg, max_iter=1000, estimator="yohoo", datasets=None, base_seed=0, results_log=None):
    """The sharded counterpart of Evaluator_PartI/II.eval (tests/evaluator.py:75-101,146-173): run every pair of the
    test set over the initialised process group (one rank per GPU), write npz / pre.log on rank 0 and return the
    Registration Recall there (None on the other ranks).  FCGF group features and keypoints are read from the
    reference's cache layout; descriptors, matches and hypotheses never touch the disk."""
    import torch
    from . import hip, RR_cal
    from .dataset import get_dataset
    if datasets is None:
        datasets = get_dataset(cfg, False)
    rank, world, local = ydist.init_from_env()
    ctx = hip.get_context(so3_dir=getattr(cfg, "SO3_related_files", None))
    load_and_broadcast_weights(cfg, ctx, need_partII=(estimator == "yohoo"))
    runner = ScenePairRunner(cfg, ctx, estimator=estimator, max_iter=max_iter, base_seed=base_seed)
    results = run_sharded(datasets, runner.run_pair, rank=rank, world=world, scene_fn=runner.setup_scene)
    torch.cuda.synchronize()
    if rank != 0:
        ydist.barrier()
        return None
    sign = 'YOHO_O' if estimator == "yohoo" else 'YOHO_C'
    for key, ds in scene_items(datasets):
        write_scene_results(cfg, ds, results[key], sign, max_iter)
    rr, flags, errors = RR_cal.benchmark(cfg, datasets, max_iter, yoho_sign=sign)
    if results_log:
        if os.path.dirname(results_log):
            os.makedirs(os.path.dirname(results_log), exist_ok=True)
        with open(results_log, 'a') as f:
            f.write(f"{datasets['wholesetname']}-{estimator}-{max_iter}iterations-{world}ranks\nMean_Registration_Recall {rr}\n\n")
    ydist.barrier()
    return rr
