"""L2 hit rate per kernel (and grid size) from a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum [TCC_EA0_RDREQ_sum] csv run.
usage: pmc_l2_report.py <dir with *_counter_collection.csv>   (hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), MI355X_MICROARCH.md)"""
import sys, glob, csv, collections, re


def main(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print("no counter csv under", d)
        return
    disp = collections.OrderedDict()
    for r in rows:
        key = (r.get("Dispatch_Id") or r.get("Dispatch_ID"), r["Kernel_Name"])
        e = disp.setdefault(key, {"name": r["Kernel_Name"], "grid": int(r.get("Grid_Size", 0) or 0), "t0": int(r["Start_Timestamp"]),
                                  "t1": int(r["End_Timestamp"]), "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = collections.OrderedDict()
    for e in disp.values():
        nm = re.sub(r"\(.*", "", e["name"])
        a = agg.setdefault((nm, e["grid"]), {"n": 0, "us": 0.0, "c": collections.Counter()})
        a["n"] += 1
        a["us"] += (e["t1"] - e["t0"]) / 1e3
        for k, v in e["c"].items():
            a["c"][k] += v
    print("| kernel | grid | launches | avg us | L2 hits / launch | L2 misses / launch | L2 hit rate | fabric read requests / launch |")
    print("|---|---|---|---|---|---|---|---|")
    tot = collections.Counter()
    for (nm, grid), a in agg.items():
        n = a["n"]
        us = a["us"] / n
        h, m, ea = a["c"].get("TCC_HIT_sum", 0) / n, a["c"].get("TCC_MISS_sum", 0) / n, a["c"].get("TCC_EA0_RDREQ_sum", 0) / n
        if "spconv" in nm:
            tot["h"] += a["c"].get("TCC_HIT_sum", 0)
            tot["m"] += a["c"].get("TCC_MISS_sum", 0)
        if us < 20 or "spconv" not in nm and "build_map" not in nm and "conv1" not in nm:
            continue
        print("| `%s` | %d | %d | %.0f | %.4g | %.4g | %.3f | %.4g |" % (nm[:52], grid, n, us, h, m, h / (h + m) if h + m else 0.0, ea))
    if tot["h"] + tot["m"]:
        print("\nall sparse-convolution launches together: L2 hit rate %.3f (%.4g hits, %.4g misses)" % (tot["h"] / (tot["h"] + tot["m"]), tot["h"], tot["m"]))


main(sys.argv[1])
