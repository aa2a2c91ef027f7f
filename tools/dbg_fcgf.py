import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
import fcgf_oracle as fo
from yoho_amd import hip, synth, weights as W
fsd = W.synth_state_dict(W.FCGF_SPEC, 3)
c = hip.Context(); c.load_fcgf(fsd)
for n, seed in ((1500, 1), (6000, 2)):
    pc = synth.surface_cloud(n, seed=seed)
    _, coords = fo.voxelize(pc, 0.025)
    F0 = fo.extract_features(pc, 0.025, fsd, normalize_feature=True)[1]
    F = c.fcgf_forward(torch.from_numpy(coords).cuda()).cpu().numpy()
    e = np.abs(F - F0).max(1)
    print(os.environ.get("YOHO_FCGF"), n, "rel", np.abs(F - F0).max() / np.abs(F0).max(), "bad rows", int((e > 1e-4).sum()), "of", len(e), "first bad", np.nonzero(e > 1e-4)[0][:10])
