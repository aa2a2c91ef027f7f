#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a per-kernel stats table (text/markdown).

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
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
    # per-launch listing for the group-conv kernel (the 4 PartI layers differ in shape)
    print("\nPer-launch durations of gconv_kernel<15,false> by grid size (one PartI pass = 4 launches):\n")
    print("| grid (threads) | calls | avg us |")
    print("|---|---|---|")
    for g, n, a in db.execute("select grid_x, count(*), avg(duration) from kernels where name like '%gconv_kernel<15%' "
                              "group by grid_x order by grid_x"):
        print(f"| {g} | {n} | {a / 1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
