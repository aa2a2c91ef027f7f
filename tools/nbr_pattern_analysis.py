"""How sparse is a sparse convolution in its KERNEL dimension, tile by tile?  (CPU, numpy; profiles/r05_nbr_sort.md)

For the benchmarked cloud (synth.surface_cloud(300000, seed 1, extent 3 m), voxel 0.025, one rotated copy) and every level of the
backbone: the average number of the 27 kernel offsets a row has a neighbour at, and - for several orders of the rows - the number of
offsets that at least one row of a 128-row workgroup / 32-row wave tile reaches, i.e. the (offset, tile) steps csrc/sparse.hip's fine-level
kernels cannot skip.  Orders: the rank / brick order the library uses, a global sort by the 27-bit neighbour pattern, by an 18-bit key
(6 face + 12 edge neighbours), and the same inside segments of S consecutive brick-order rows.

    python tools/nbr_pattern_analysis.py [group element = 7]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import fcgf_oracle as fo            # noqa: E402  (a measurement tool may use the checker)
from yoho_amd import synth           # noqa: E402
from yoho_amd.tables import default_tables   # noqa: E402


def level_masks(c, ts):
    key = lambda a: (a[:, 0].astype(np.int64) + 2 ** 20) * 2 ** 42 + (a[:, 1].astype(np.int64) + 2 ** 20) * 2 ** 21 + (a[:, 2].astype(np.int64) + 2 ** 20)
    k = key(c)
    ks = np.sort(k)
    m = np.zeros(len(c), np.int64)
    for i, o in enumerate(fo.kernel_offsets(3, ts)):
        q = key(c + o)
        pos = np.searchsorted(ks, q)
        pos[pos >= len(ks)] = 0
        m |= (ks[pos] == q).astype(np.int64) << i
    return m


def brick_order(c, ts):
    o = (c.min(0) // 16) * 16
    X, Y, Z = ((c - o) // ts).T
    return np.lexsort((X & 31, Y & 7, Z & 7, X >> 5, Y >> 3, Z >> 3))


def steps(m, rows):
    pad = (-len(m)) % rows
    u = np.bitwise_or.reduce(np.concatenate([m, np.zeros(pad, np.int64)]).reshape(-1, rows), axis=1)
    return sum(bin(int(v)).count("1") for v in u) / len(u)


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    pc = synth.surface_cloud(300000, seed=1, extent=3.0) @ default_tables().R64[g].T
    c = fo.voxelize(pc, 0.025)[1]
    idx = lambda dx, dy, dz: (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1)
    faces = [idx(1, 0, 0), idx(-1, 0, 0), idx(0, 1, 0), idx(0, -1, 0), idx(0, 0, 1), idx(0, 0, -1)]
    edges = [idx(a, b, 0) for a in (-1, 1) for b in (-1, 1)] + [idx(a, 0, b) for a in (-1, 1) for b in (-1, 1)] + [idx(0, a, b) for a in (-1, 1) for b in (-1, 1)]
    print("| level (stride) | rows | neighbours per row | order | offsets per 128-row tile | per 32-row tile |")
    print("|---|---|---|---|---|---|")
    cl = c
    for ts in (1, 2, 4, 8):
        if ts > 1:
            cl = fo.stride_coords(cl, ts)
        m = level_masks(cl, ts)
        n = len(m)
        avg = float(np.mean([bin(int(v)).count("1") for v in m[:50000]]))
        key18 = np.zeros(n, np.int64)
        for j, b in enumerate(faces + edges):
            key18 |= ((m >> b) & 1) << (17 - j)
        bo = brick_order(cl, ts)
        mb, kb = m[bo], key18[bo]
        rows = [("brick order (shipped)", mb), ("global sort by the 27-bit pattern", np.sort(m)), ("global sort by the 18-bit key", m[np.argsort(key18, kind="stable")])]
        for S in (2048, 8192, 32768):
            rows.append((f"18-bit key inside segments of {S} brick-order rows", mb[np.lexsort((kb, np.arange(n) // S))]))
        for name, mm in rows:
            print(f"| {ts} | {n} | {avg:.1f} | {name} | {steps(mm, 128):.1f} | {steps(mm, 32):.1f} |")


if __name__ == "__main__":
    main()
