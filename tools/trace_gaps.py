#!/usr/bin/env python
"""Idle gaps of a rocprofv3 rocpd kernel trace: device busy time against the span of the last `frac` of the trace, and the
largest gaps with the kernels on either side (where the host makes the device wait).

    python tools/trace_gaps.py <dir or .db> [frac=0.33] [top=25]
"""
import glob, os, sqlite3, sys


def main(path, frac=0.33, top=25):
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = db.execute(f"select name, {s}, {e} from kernels order by {s}").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    cut = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[1] >= cut]
    span = max(r[2] for r in rows) - rows[0][1]
    busy, cur_end, gaps = 0, rows[0][1], []
    for i, (nm, a, b) in enumerate(rows):
        if a > cur_end:
            gaps.append((a - cur_end, rows[i - 1][0], nm))
        busy += max(0, b - max(a, cur_end))
        cur_end = max(cur_end, b)
    print(f"last {frac:.2f} of the trace: span {span / 1e6:.2f} ms, device busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps, {len(rows)} launches")
    hist = [(50e3, 0, 0), (20e3, 0, 0), (10e3, 0, 0), (5e3, 0, 0), (0, 0, 0)]
    for lim in (100e3, 50e3, 20e3, 10e3, 5e3, 0):
        sel = [g[0] for g in gaps if g[0] >= lim]
        print(f"  gaps >= {lim / 1e3:.0f} us: {len(sel)}, {sum(sel) / 1e6:.2f} ms")
    print("\n| gap us | after | before |\n|---|---|---|")
    for g, a, b in sorted(gaps, reverse=True)[:top]:
        print(f"| {g / 1e3:.1f} | `{a[:60]}` | `{b[:60]}` |")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.33, int(sys.argv[3]) if len(sys.argv) > 3 else 25)
