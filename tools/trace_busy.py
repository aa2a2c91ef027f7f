#!/usr/bin/env python
"""Device busy fraction in sliding windows of a rocprofv3 rocpd kernel trace (union of kernel intervals over all streams):
    python tools/trace_busy.py <dir or .db> [window_ms=50]
prints the busiest windows - e.g. the timed steps of bench.py, where two pairs are in flight on two streams."""
import glob, os, sqlite3, sys


def main(path, win_ms=50.0):
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = db.execute(f"select {s}, {e} from kernels order by {s}").fetchall()
    # union of intervals
    merged = []
    for a, b in rows:
        if merged and a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    t0, t1 = merged[0][0], merged[-1][1]
    win = win_ms * 1e6
    out = []
    w0 = t0
    while w0 + win <= t1:
        busy = sum(max(0, min(b, w0 + win) - max(a, w0)) for a, b in merged if b > w0 and a < w0 + win)
        out.append((busy / win, (w0 - t0) / 1e6))
        w0 += win / 2
    out.sort(reverse=True)
    print(f"{len(rows)} launches over {(t1 - t0) / 1e6:.0f} ms; busiest {win_ms:.0f} ms windows (busy fraction @ offset ms):")
    print("  " + "  ".join(f"{f:.3f}@{o:.0f}" for f, o in out[:12]))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 50.0)
