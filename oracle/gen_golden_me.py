"""tests/golden/me_maps.npz - coordinate maps and kernel maps of the FCGF backbone's convolutions as the REAL MinkowskiEngine CPU
coordinate manager produces them (oracle/_ref/me_maps.so, built by oracle/build_me_ref.py from the reference's own
src/coordinate_map_manager.cpp; build container only).  tests/test_fcgf_oracle.py holds oracle/fcgf_oracle.py's restatement
(stride_coords, kernel_map incl. transpose) against these vectors, so the part of the backbone oracle that used to rest on source
reading alone - strided and transposed kernel maps - is pinned by the reference itself.

A kernel map is stored as its set of (kernel index, input coordinate, output coordinate) triples, sorted: MinkowskiEngine numbers
the rows of a strided map in hash-table order, the oracle in first-occurrence order; row numbering of an intermediate level is
internal (every output row receives at most one input per kernel index, and the sum runs over kernel indices in order), the
triples are what defines the convolution.  Large maps are stored as (count, sha256 of the sorted int32 triple array).

    python oracle/gen_golden_me.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import build_me_ref  # noqa: E402
from yoho_amd import synth  # noqa: E402

FULL_LIMIT = 4000          # triples stored in full up to this many, digest beyond


def triples(km, cin, cout):
    rows = []
    for k, (a, b) in enumerate(km):
        a, b = a.numpy(), b.numpy()
        if len(a):
            rows.append(np.concatenate([np.full((len(a), 1), k, np.int32), cin[a], cout[b]], 1))
    t = np.concatenate(rows).astype(np.int32) if rows else np.zeros((0, 7), np.int32)
    return t[np.lexsort(t.T[::-1])]


def digest(t):
    return hashlib.sha256(np.ascontiguousarray(t, dtype=np.int32).tobytes()).hexdigest()


def cases():
    rs = np.random.RandomState(11)
    pc = synth.surface_cloud(900, seed=5)
    q = np.floor(pc / 0.025).astype(np.int32)
    q = np.concatenate([q, q[:40]])                       # duplicate voxels: first occurrence wins
    q[:, 0] -= 17                                         # negative coordinates
    yield "surface", q
    yield "random", rs.randint(-9, 9, size=(500, 3)).astype(np.int32)
    yield "line", np.array([[v, 0, 0] for v in range(-7, 8)], dtype=np.int32)


def main():
    build_me_ref.build()
    me = build_me_ref.load()
    assert me is not None, "oracle/_ref/me_maps.so is missing and /root/reference is not here to build it"
    out = {}
    for name, q in cases():
        coords = np.ascontiguousarray(np.concatenate([np.zeros((len(q), 1), np.int32), q], 1))
        o = me.fcgf_maps(torch.from_numpy(coords), 7)
        out[f"{name}_input"] = q
        out[f"{name}_unique_map"] = o["unique_map"].numpy().astype(np.int64)
        c = [o[f"coords_{l}"].numpy()[:, 1:].astype(np.int32) for l in range(4)]
        for l in range(4):
            out[f"{name}_coords_{l}"] = c[l]              # rows in the reference's own order (level 0: insertion order)
        maps = {"conv1": (o["conv1"], c[0], c[0])}
        for l in range(4):
            maps[f"s1_{l}"] = (o[f"conv_s1_{l}"], c[l], c[l])
        for l in range(3):
            maps[f"s2_{l}"] = (o[f"conv_s2_{l}"], c[l], c[l + 1])
        for l in (3, 2, 1):
            tc = o[f"tr_out_coords_{l}"].numpy()[:, 1:].astype(np.int32)
            assert bool(o[f"tr_out_is_level_{l}"]) and np.array_equal(tc, c[l - 1])     # the transposed conv lands on the existing finer map
            maps[f"tr_{l}"] = (o[f"conv_tr_{l}"], c[l], tc)
        for mn, (km, cin, cout) in maps.items():
            t = triples(km, cin, cout)
            out[f"{name}_{mn}_count"] = np.int64(len(t))
            out[f"{name}_{mn}_sha256"] = np.array(digest(t))
            if len(t) <= FULL_LIMIT:
                out[f"{name}_{mn}_triples"] = t
    # tests/python/kernel_map.py's figure (2 batch items x 8 points, 2-D, kernel 3, stride 2): the reference itself yields 26 pairs,
    # not the 16 its stale python test asserts
    fig = ["   X   ", "  X X  ", " XXXXX "]
    pts = np.array([[0, i, j] for i, r in enumerate(fig) for j, ch in enumerate(r) if ch != " "], dtype=np.int32)
    two = np.ascontiguousarray(np.concatenate([pts, pts + np.array([1, 0, 0], np.int32)]))
    o = me.strided_map_nd(torch.from_numpy(two))
    out["figure_points"] = pts[:, 1:]
    out["figure_pairs_total"] = np.int64(sum(len(a) for a, _ in o["map"]))
    out["figure_out_coords_item0"] = np.array(sorted(tuple(r[1:]) for r in o["out_coords"].numpy().tolist() if r[0] == 0), dtype=np.int32)
    path = os.path.join(REPO, "tests", "golden", "me_maps.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", "figure pairs", int(out["figure_pairs_total"]))


if __name__ == "__main__":
    main()
