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
