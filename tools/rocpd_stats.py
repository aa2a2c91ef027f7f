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
    # bench.py's roofline reads the GEMM launches of the LAST of nine back-to-back PartI passes, three times (27 passes at the end of
    # the run when the dataset / raw-cloud / CPU legs are off): the same launches from this trace, so that the line can be recomputed
    # from the file - the table above averages every launch of the run, those of the timed steps (pipeline clocks) included
    try:
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        tcol = "start" if "start" in cols else ("start_time" if "start_time" in cols else None)
        if tcol:
            print("\nThe GEMM / transform launches of the last 27 PartI passes (bench.py's profiled passes: 3 x 9 back to back), by launch order in a pass; "
                  "'every ninth' = the launches bench.py's roofline reads:\n")
            print("| kernel | grid (threads) | launches per pass | avg us, all 27 passes | avg us, every ninth pass (9th, 18th, 27th) |")
            print("|---|---|---|---|---|")
            tot_all = tot_9 = 0.0
            for pat, per in (("fgemm3_kernel", None), ("fgemm3s_kernel", None), ("gft16x_kernel", None)):
                grids = [g for (g,) in db.execute(f"select distinct grid_x from kernels where name like '%{pat}%'")]
                for g in sorted(grids):
                    rows = db.execute(f"select duration from kernels where name like '%{pat}%' and grid_x = ? order by {tcol}", (g,)).fetchall()
                    npass = 27
                    # launches per pass: fgemm3 runs 32 -> 256 and 512 -> 256 on the smaller of its two grids, 256 -> 512 on the larger
                    k = 3 if pat == "gft16x_kernel" else (2 if (pat == "fgemm3_kernel" and len(grids) > 1 and g == min(grids)) else 1)
                    tail = [r[0] for r in rows[-npass * k:]]
                    if len(tail) < npass * k:
                        continue
                    ninth = [v for i, v in enumerate(tail) if (i // k) % 9 == 8]
                    a_all, a_9 = sum(tail) / len(tail) / 1e3, sum(ninth) / len(ninth) / 1e3
                    print(f"| `{pat}` | {g} | {k} | {a_all:.1f} | {a_9:.1f} |")
                    if pat != "gft16x_kernel":
                        tot_all += a_all * k
                        tot_9 += a_9 * k
            if tot_9 > 0:
                print(f"\nsum over the four GEMM launches of a pass: {tot_all / 1e3:.4f} ms (all 27), {tot_9 / 1e3:.4f} ms (every ninth) -> 4.078 TFLOP / that / 2500 TFLOP/s = "
                      f"{4.078 / (tot_all / 1e3) / 2500 * 1e3:.4f} / {4.078 / (tot_9 / 1e3) / 2500 * 1e3:.4f} of the fp16 peak")
    except Exception as e:
        print("\n(no tail table:", e, ")")


if __name__ == "__main__":
    main(sys.argv[1])
