"""CPU tests of the N > 1 path with world_size 2 on the gloo backend: the one-time checkpoint
broadcast, the pair sharding (no data-path collective) and the result gather / max-over-ranks timing."""
import os
import socket
import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from yoho_amd import dist as ydist, weights as W
    r, w, _ = ydist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    sd = W.synth_state_dict(W.PARTI_SPEC, 7) if rank == 0 else None
    sd = ydist.broadcast_state_dict(sd, W.PARTI_SPEC, src=0, device=torch.device("cpu"))
    ref = W.synth_state_dict(W.PARTI_SPEC, 7)
    ok = all(np.array_equal(sd[k], ref[k]) for k, _ in W.PARTI_SPEC)
    pairs = [(i, i + 1) for i in range(7)]
    mine = ydist.shard(pairs, rank, world)
    allres = ydist.gather_results([(p, rank) for p in mine])
    flat = sorted(x for part in allres for x in part)
    t = ydist.max_over_ranks(1.0 + rank, device=torch.device("cpu"))
    ydist.barrier()
    q.put((rank, ok, mine, flat, t))
    torch.distributed.destroy_process_group()


def test_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pairs = [(i, i + 1) for i in range(7)]
    assert res[0][1] and res[1][1]                      # every rank holds rank 0's checkpoint bit-exactly
    assert res[0][2] == pairs[0::2] and res[1][2] == pairs[1::2]
    assert sorted(p for p, _ in res[0][3]) == pairs     # every pair processed exactly once
    assert res[0][3] == res[1][3]
    assert res[0][4] == res[1][4] == 2.0                # max over ranks


def test_single_process_noops():
    from yoho_amd import dist as ydist, weights as W
    sd = W.synth_state_dict(W.PARTII_SPEC, 1)
    assert ydist.broadcast_state_dict(sd, W.PARTII_SPEC) is sd
    assert ydist.shard(list(range(5)), 0, 1) == list(range(5))
    assert ydist.max_over_ranks(3.5) == 3.5 and ydist.gather_results([1]) == [[1]]


def _launched_world1(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from yoho_amd import dist as ydist, weights as W
    r, w, _ = ydist.init_from_env("gloo")
    sd = W.synth_state_dict(W.PARTII_SPEC, 3)
    got = ydist.broadcast_state_dict(sd, W.PARTII_SPEC, device=torch.device("cpu"))
    same = all(np.array_equal(got[k], np.asarray(v, np.float32)) for k, v in sd.items() if not k.endswith("num_batches_tracked"))
    out = (r, w, ydist.active(), torch.distributed.is_initialized(), got is not sd, same,
           ydist.all_ranks(1.25, device=torch.device("cpu")), ydist.max_over_ranks(2.5, device=torch.device("cpu")),
           ydist.gather_results({"a": 1}))
    ydist.barrier()
    q.put(out)
    torch.distributed.destroy_process_group()


def test_launched_with_one_rank_still_goes_through_the_backend():
    """Under a launcher (RANK / WORLD_SIZE / MASTER_* set) a one-rank job creates its process group and its helpers call the backend:
    the code between `torch.distributed.run` and the first timed step is the same at N = 1 and N = 8 (tests/test_gpu_rccl.py runs
    this with "nccl" on the GPU box)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_launched_world1, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=120)
    p.join(30)
    assert p.exitcode == 0
    assert out == (0, 1, True, True, True, True, [1.25], 2.5, [{"a": 1}])


# ---- dataset-level driver (yoho_amd/run_dataset.py): plan, ordering, completeness ------------------------------------
class _FakeScene:
    def __init__(self, name, nfrag, npairs, seed):
        rs = np.random.RandomState(seed)
        self.name = name
        self.pc_ids = [str(i) for i in range(nfrag)]
        allp = [(str(a), str(b)) for a in range(nfrag) for b in range(a + 1, nfrag)]
        self.pair_ids = [allp[i] for i in sorted(rs.permutation(len(allp))[:npairs])]


def _fake_datasets():
    # sizes shaped like 3DMatch (utils/dataset.py:163-167): one scene much larger than the per-rank share
    ds = {"wholesetname": "fake"}
    for i, (nf, npair) in enumerate([(20, 150), (6, 9), (5, 7), (7, 11), (4, 3)]):
        ds[f"scene{i}"] = _FakeScene(f"fake/scene{i}", nf, npair, i)
    return ds


def _fake_pair_fn(calls):
    from yoho_amd import run_dataset as rd

    def fn(dataset, pair):
        calls.append((dataset.name, pair))
        seed = rd.pair_seed(5, dataset.name, *pair)
        rs = np.random.RandomState(seed & 0xFFFFFFFF)
        T = np.concatenate([np.eye(3), rs.rand(3, 1)], 1)
        return {"trans": T, "recalltime": int(seed % 1000), "rank": int(os.environ.get("RANK", "0"))}
    return fn


# pairs per scene of the 3DMatch test set as its gt.log files list them (1623 in all, the benchmark's well-known total; 3DLoMatch's
# gtLo.log lists 1781 over the same fragments), fragments per scene from utils/dataset.py:163-167
_3DMATCH_PAIRS = {"kitchen": 506, "home1": 156, "home2": 208, "hotel1": 226, "hotel2": 104, "hotel3": 54, "study": 292, "lab": 77}
_3DMATCH_FRAGS = {"kitchen": 60, "home1": 60, "home2": 60, "hotel1": 55, "hotel2": 57, "hotel3": 37, "study": 66, "lab": 38}
# ETH (utils/dataset.py:195-196) and WHU-TLS (:207-208): the gt.log files are not in the tree.  ASSUMPTION: ETH lists every
# fragment pair of a scene (C(n,2): an upper bound on its overlapping pairs), WHU-TLS the n - 1 consecutive scan pairs
# (utils/RR_cal.py treats it as the consecutive-pairs benchmark)
_ETH_FRAGS = {"gazebo_summer": 32, "gazebo_winter": 31, "wood_autumn": 32, "wood_summer": 37}
_ETH_PAIRS = {k: n * (n - 1) // 2 for k, n in _ETH_FRAGS.items()}
_WHU_FRAGS = {"Park": 32, "Mountain": 6, "Campus": 10, "RiverBank": 7, "UndergroundExcavation": 12, "Tunnel": 7}
_WHU_PAIRS = {k: n - 1 for k, n in _WHU_FRAGS.items()}


def test_plan_shards_properties():
    from yoho_amd import run_dataset as rd
    sizes, frags = _3DMATCH_PAIRS, _3DMATCH_FRAGS
    for world in (1, 2, 3, 8, 16):
        for fr in (None, frags):
            plan = rd.plan_shards(sizes, world, fr)
            assert len(plan) == world
            seen = sorted((s, p) for part in plan for s, pos in part for p in pos)
            assert seen == sorted((s, p) for s, n in sizes.items() for p in range(n))          # each pair exactly once
            # parts are contiguous blocks of a scene's pair list and no part is a sliver
            for part in plan:
                for s, pos in part:
                    assert pos == list(range(pos[0], pos[0] + len(pos)))
                    assert len(pos) >= min(rd.MIN_PART, sizes[s])
    assert rd.plan_shards(sizes, 8, frags) == rd.plan_shards(dict(reversed(list(sizes.items()))), 8, frags)       # independent of dict order
    assert rd.plan_shards({}, 4) == [[], [], [], []] and rd.plan_shards({"a": 0, "b": 3}, 2) in ([[("b", [0, 1, 2])], []], [[], [("b", [0, 1, 2])]])


def test_plan_balance_on_real_scene_sizes():
    """VERDICT r2 #8a: planned imbalance (max / mean rank load under run_dataset.part_cost: pairs + FRAG_COST per fragment a part has
    to load and describe) at the world sizes the scaling bench uses.  3DMatch / 3DLoMatch and ETH stay within 10 % at every world
    size up to 8; WHU-TLS has 68 pairs in 6 scenes - too little work to balance over 8 ranks (parts below MIN_PART pairs are not
    cut), it is asserted up to 4 ranks."""
    from yoho_amd import run_dataset as rd
    lo_pairs = {k: round(n * 1781 / 1623) for k, n in _3DMATCH_PAIRS.items()}            # 3DLoMatch: same fragments, 1781 pairs
    for name, pairs, frags, worlds in (("3dmatch", _3DMATCH_PAIRS, _3DMATCH_FRAGS, (2, 8)), ("3dLomatch", lo_pairs, _3DMATCH_FRAGS, (2, 8)),
                                       ("ETH", _ETH_PAIRS, _ETH_FRAGS, (2, 4, 8)), ("WHU-TLS", _WHU_PAIRS, _WHU_FRAGS, (2, 4))):
        for world in worlds:
            for fc in (0.0, rd.FRAG_COST, 2 * rd.FRAG_COST):          # pairs only, the default model, fragments twice as dear
                plan = rd.plan_shards(pairs, world, frags, fc)
                loads = rd.plan_loads(plan, frags, fc)
                ratio = max(loads) / (sum(loads) / world)
                assert ratio <= (1.10 if name != "WHU-TLS" else 1.18), (name, world, fc, ratio, loads)
                # cutting scenes must not cost more than a quarter of the work it spreads
                whole = sum(rd.part_cost(n, frags[k], fc) for k, n in pairs.items())
                assert sum(loads) <= 1.6 * whole, (name, world, fc, sum(loads), whole)


def _driver_worker(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from yoho_amd import dist as ydist, run_dataset as rd
    ydist.init_from_env("gloo")
    ds = _fake_datasets()
    calls, setups = [], []
    res = rd.run_sharded(ds, _fake_pair_fn(calls), rank=rank, world=world, scene_fn=lambda d, pairs: setups.append((d.name, len(pairs))))
    if rank == 0:
        class Cfg:
            output_cache_fn = tmp
        for key, d in rd.scene_items(ds):
            rd.write_scene_results(Cfg, d, res[key], "YOHO_O", 1000)
    ydist.barrier()
    q.put((rank, calls, setups, {k: [(r["recalltime"], r["rank"]) for r in v] for k, v in res.items()}))
    torch.distributed.destroy_process_group()


def test_sharded_driver_world2_matches_world1(tmp_path):
    from yoho_amd import run_dataset as rd
    from yoho_amd.estimator import format_log_entry
    ds = _fake_datasets()
    calls1 = []
    ref = rd.run_sharded(ds, _fake_pair_fn(calls1), rank=0, world=1)
    assert [c for c in calls1] == [(d.name, tuple(p)) for _, d in sorted(rd.scene_items(ds), key=lambda kd: -len(kd[1].pair_ids))
                                   for p in d.pair_ids]
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # completeness: every pair ran on exactly one rank; both ranks hold the full, identically ordered result
    ran = sorted(c for _, calls, _, _ in out for c in calls)
    assert ran == sorted((d.name, tuple(p)) for _, d in rd.scene_items(ds) for p in d.pair_ids)
    assert out[0][3] == out[1][3]
    for key, d in rd.scene_items(ds):
        assert [t for t, _ in out[0][3][key]] == [r["recalltime"] for r in ref[key]]      # same numbers as the 1-rank run
    # the big scene was cut over both ranks, the small ones were not; descriptors are set up once per (rank, scene part)
    ranks_of = {key: {rk for _, rk in out[0][3][key]} for key in out[0][3]}
    assert ranks_of["scene0"] == {0, 1} and sum(len(v) for v in ranks_of.values()) <= len(ranks_of) + 2
    assert sorted(s for _, _, setups, _ in out for s, _ in setups).count("fake/scene0") == 2
    # rank 0 wrote pre.log in dataset.pair_ids order with the gathered transforms
    for key, d in rd.scene_items(ds):
        text = open(f"{tmp_path}/Testset/{d.name}/Match/YOHO_O/1000iters/pre.log").read()
        assert text == "".join(format_log_entry(a, b, len(d.pc_ids), r["trans"]) for (a, b), r in zip(d.pair_ids, ref[key]))
        z = np.load(f"{tmp_path}/Testset/{d.name}/Match/YOHO_O/1000iters/{d.pair_ids[-1][0]}-{d.pair_ids[-1][1]}.npz")
        assert np.array_equal(z["trans"], ref[key][-1]["trans"]) and int(z["recalltime"]) == ref[key][-1]["recalltime"]


def _failing_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from yoho_amd import dist as ydist, run_dataset as rd
    ydist.init_from_env("gloo")
    ds = _fake_datasets()

    def pair_fn(dataset, pair):
        if rank == 1:
            raise ValueError("boom on rank 1")
        return {"trans": np.eye(4)[:3], "recalltime": 0}
    try:
        rd.run_sharded(ds, pair_fn, rank=rank, world=world)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    ydist.barrier()
    torch.distributed.destroy_process_group()


def test_sharded_driver_failure_on_one_rank_reaches_every_rank():
    """a rank whose pairs fail still takes part in the gather, so no rank is left waiting: every rank raises the same error"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all("boom on rank 1" in msg and "rank 1: ValueError" in msg for _, msg in out), out
