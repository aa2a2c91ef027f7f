"""Dataset-scale backbone driver (YOHO_testset.testset_create.batch_feature_extraction): fragments/s from point clouds in memory to
FCGF_Input_Group_feature/*.npy on disk, with one and two backbone lanes.   usage: bench_testset.py [fragments] [points] [keypoints]"""
import sys, os, time, types, tempfile, shutil, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import synth, weights as W
from yoho_amd.YOHO_testset import testset_create

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
nk = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7}, "state_dict": fsd}
clouds = [synth.surface_cloud(n, seed=1 + i, extent=3.0) for i in range(3)]
rs = np.random.RandomState(0)
kps = [c[rs.permutation(len(c))[:nk]] for c in clouds]


class DS:
    pc_ids = [str(i) for i in range(nf)]
    get_pc = staticmethod(lambda i: clouds[int(i) % 3])
    get_kps = staticmethod(lambda i: kps[int(i) % 3])


digest = {}
for rep, lanes in enumerate([int(v) for v in os.environ.get("BENCH_TESTSET_LANES", "2,1,2,1").split(",")]):
    tmp = tempfile.mkdtemp(prefix="yoho_testset_")
    cfg = types.SimpleNamespace(model=ck, voxel_size=0.025, dataset="synth", output_dir=tmp, origin_dir=tmp, datasets={"wholesetname": "synth", "room": DS()})
    os.environ["YOHO_FCGF_LANES"] = str(lanes)
    tc = testset_create(cfg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tc.batch_feature_extraction()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    h = hashlib.sha256()
    for i in range(nf):
        h.update(np.load(f"{tmp}/Testset/synth/room/FCGF_Input_Group_feature/{i}.npy").tobytes())
    digest.setdefault(lanes, set()).add(h.hexdigest())
    shutil.rmtree(tmp)
    print(f"rep {rep}: lanes {lanes}: {nf} fragments ({n} points, {nk} keypoints) in {dt:.3f} s = {dt / nf * 1e3:.1f} ms per fragment, {nf / dt:.2f} fragments/s", flush=True)
print("files identical between the modes and repeats:", len(set().union(*digest.values())) == 1)
