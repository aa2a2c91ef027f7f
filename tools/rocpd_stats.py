#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a per-kernel stats table (text/markdown).

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/r01_kernel_stats.md
"""
import glob
import os
import sqlite3
import sys


def main(path):
    if os.path.isdir(path):
        found = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
        if not found:
            raise SystemExit("no .db under " + path)
        path = found[0]
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | lds B | vgpr | agpr | sgpr |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        print(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | "
              f"{100 * r[2] / tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")
    # per-launch listing by grid size (the layers of one pass share a kernel but differ in shape)
    print("\nPer-launch durations by kernel and grid size:\n")
    print("| kernel | grid (threads) | calls | avg us |")
    print("|---|---|---|---|")
    for nm, g, n, a in db.execute("select name, grid_x, count(*), avg(duration) from kernels group by name, grid_x "
                                  "having sum(duration) > 0 order by name, grid_x"):
        nm = nm if len(nm) < 70 else nm[:67] + "..."
        print(f"| `{nm}` | {g} | {n} | {a / 1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
