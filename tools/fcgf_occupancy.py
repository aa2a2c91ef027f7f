"""Occupancy of the 3^3 kernel region on the four levels of the FCGF backbone for the benches' synthetic surface cloud, and the
MFMA-tile ratio a per-offset compaction of (input, output) pairs over row blocks of RB rows would reach.
CPU only.  usage: python tools/fcgf_occupancy.py"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yoho_amd import synth
pc = synth.surface_cloud(300000, seed=1, extent=3.0)
c = np.unique(np.floor(pc/0.025).astype(np.int64), axis=0)
def stats(c, ts, name):
    # Msame occupancy at stride ts
    key = lambda a: (a[:,0]+(1<<20))*(1<<42) + (a[:,1]+(1<<20))*(1<<21) + (a[:,2]+(1<<20))
    ks = np.sort(key(c))
    occ=[]
    for dx in (-1,0,1):
        for dy in (-1,0,1):
            for dz in (-1,0,1):
                q = c + np.array([dx,dy,dz])*ts
                kq = key(q)
                i = np.searchsorted(ks,kq); i[i>=len(ks)] = 0
                occ.append((ks[i]==kq))
    occ=np.array(occ)  # 27 x n
    print(name, "rows", len(c), "mean neighbours %.2f of 27 = %.3f"%(occ.sum(0).mean(), occ.mean()))
    # tile stats: for blocks of RB rows in given order: tiles needed with compaction vs dense
    for RB in (32,64,128,256):
        nb = len(c)//RB
        o = occ[:, :nb*RB].reshape(27, nb, RB).sum(2)   # 27 x nb counts
        tiles = np.ceil(o/32).sum()
        dense = 27*nb*RB/32
        print("   RB=%d: compact tiles / dense tiles = %.3f ; lockstep4 rounds(ceil(cnt/128)) %.3f" % (RB, tiles/dense, (np.ceil(o/128).sum()*4)/dense if RB==256 else 0))
    return occ
rs=np.random.RandomState(0)
c0 = c[rs.permutation(len(c))]
for l in range(4):
    ts = 1<<l
    cl = np.unique((c0//ts)*ts, axis=0) if l else c0
    cl = cl[rs.permutation(len(cl))]
    stats(cl, ts, "L%d"%l)
