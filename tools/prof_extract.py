import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yoho_amd import synth, weights as W, hip
from yoho_amd.fcgf_feat import fcgf_extractor
n=int(sys.argv[1]) if len(sys.argv)>1 else 300000
fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
ck = {"config": {"model": "ResUNetBN2C", "model_n_out": 32, "normalize_feature": True, "conv1_kernel_size": 7}, "state_dict": fsd}
ctx = hip.get_context()
fx = fcgf_extractor(ck, ctx=ctx)
pc = synth.surface_cloud(n, seed=1, extent=3.0)
pc_d = torch.from_numpy(pc).cuda()
kp_d = pc_d[:5000].contiguous()
R = ctx.tables.R64
acc = {}
def lap(k, t0):
    torch.cuda.synchronize(); t=time.perf_counter(); acc[k]=acc.get(k,0)+t-t0; return t
out = torch.empty((5000,32,60), device="cuda")
for rep in range(2):
    acc.clear()
    torch.cuda.synchronize(); t=time.perf_counter()
    for g0 in range(0,60,6):
        Rts=[torch.from_numpy(np.ascontiguousarray(R[g].T)).cuda() for g in range(g0,g0+6)]
        pcs=[pc_d@Rt for Rt in Rts]
        t=lap("rotate",t)
        vox=[ctx.fcgf_voxelize(p,0.025) for p in pcs]
        t=lap("voxelize",t)
        feats=ctx.fcgf_forward_batch([c for _,c in vox])
        t=lap("backbone",t)
        for j in range(6):
            q=(kp_d@Rts[j]).float().contiguous()
            ds=pcs[j][vox[j][0]].float().contiguous()
            t=lap("gather pts",t)
            _,idx=ctx.nn_search(q,ds,want_dist=False,squared=True)
            t=lap("nn",t)
            out[:,:,g0+j]=feats[j][idx]
            t=lap("scatter",t)
print({k: round(v*1e3,1) for k,v in acc.items()}, "total", round(sum(acc.values())*1e3,1), "voxels", vox[0][1].shape[0])
