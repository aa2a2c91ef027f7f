#!/usr/bin/env python
"""Per-stream timeline of the LAST yoho_extractor.run of a rocprofv3 rocpd kernel trace (tools/bench_extract.py under
`rocprofv3 --kernel-trace`): span, device busy (union over streams), busy per stream, time by kernel family, and a coarse
strip chart (one character per `bin_us`) per stream.   python tools/trace_lanes.py <dir or .db> [bin_us=250] [last_ms: the last so many ms of the trace instead]"""
import glob, os, sqlite3, sys


def fam(n):
    for k, v in (("spconv", "C"), ("conv1_", "C"), ("vox_", "v"), ("rk_", "m"), ("build_map", "k"), ("invert_map", "k"), ("parity", "k"), ("bbox", "m"),
                 ("block_scan", "m"), ("gt_", "t"), ("fgemm", "P"), ("gft16", "P"), ("head16", "P"), ("finalize", "P"), ("copyBuffer", "c"), ("fillBuffer", "f")):
        if k in n:
            return v
    return "o"


def main(path, bin_us=250.0, last_ms=0.0):
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = db.execute(f"select name, {s}, {e}, {q if q else 0} from kernels order by {s}").fetchall()
    # the last fragment = from the last aabb / first vox kernel after the last finalize_partI-but-one
    fin = [i for i, r in enumerate(rows) if "finalize_partI" in r[0]]
    if last_ms > 0:                                  # the last `last_ms` of the trace instead of the last extractor call
        tend = max(r[2] for r in rows)
        rows = [r for r in rows if r[1] >= tend - last_ms * 1e6]
    else:
        lo = fin[-2] + 1 if len(fin) >= 2 else 0
        rows = rows[lo:fin[-1] + 1]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    print(f"{len(rows)} launches, span {(t1 - t0) / 1e6:.2f} ms")
    merged = []
    for _, a, b, _ in rows:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    print(f"device busy (union) {sum(b - a for a, b in merged) / 1e6:.2f} ms")
    streams = sorted({r[3] for r in rows})
    byfam = {}
    for n, a, b, st in rows:
        byfam[fam(n)] = byfam.get(fam(n), 0) + (b - a)
    print("sum of kernel durations by family (C conv, v voxelise, m coordinate maps, k kernel maps, t transfer, P PartI, c copy, f fill, o other): " +
          ", ".join(f"{k} {v / 1e6:.2f}" for k, v in sorted(byfam.items(), key=lambda x: -x[1])))
    nb = int((t1 - t0) / (bin_us * 1e3)) + 1
    for st in streams:
        rs = [r for r in rows if r[3] == st]
        print(f"stream {st}: {len(rs)} launches, busy {sum(r[2] - r[1] for r in rs) / 1e6:.2f} ms")
        strip = []
        for i in range(nb):
            w0, w1 = t0 + i * bin_us * 1e3, t0 + (i + 1) * bin_us * 1e3
            acc = {}
            for n, a, b, _ in rs:
                if b > w0 and a < w1:
                    acc[fam(n)] = acc.get(fam(n), 0) + min(b, w1) - max(a, w0)
            strip.append(max(acc, key=acc.get) if acc and max(acc.values()) > 0.2 * bin_us * 1e3 else ".")
        print("  " + "".join(strip))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 250.0, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
