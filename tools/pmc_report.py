"""Summarise a rocprofv3 --pmc ... --output-format csv run: per kernel (and grid size) duration, clock, MFMA utilisation.
usage: pmc_report.py <dir with *_counter_collection.csv>"""
import sys, glob, csv, collections, re

def main(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print("no counter csv under", d); return
    disp = collections.OrderedDict()
    for r in rows:
        key = (r.get("Dispatch_Id") or r.get("Dispatch_ID"), r["Kernel_Name"])
        e = disp.setdefault(key, {"name": r["Kernel_Name"], "grid": int(r.get("Grid_Size", 0) or 0),
                                  "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"]), "c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = collections.OrderedDict()
    for e in disp.values():
        nm = re.sub(r"\(.*", "", e["name"])
        a = agg.setdefault((nm, e["grid"]), {"n": 0, "us": 0.0, "c": collections.Counter()})
        a["n"] += 1; a["us"] += (e["t1"] - e["t0"]) / 1e3
        for k, v in e["c"].items(): a["c"][k] += v
    names = sorted({k for a in agg.values() for k in a["c"]})
    print("| kernel | grid | n | us | " + " | ".join(names) + " | clock GHz | MFMA util |")
    print("|---|---|---|---|" + "---|" * (len(names) + 2))
    for (nm, grid), a in agg.items():
        n = a["n"]; us = a["us"] / n
        c = {k: v / n for k, v in a["c"].items()}
        clock = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (us * 1e3) if us else 0
        act = c.get("GRBM_GUI_ACTIVE", 0) / 8
        util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * act) if act else 0
        if us < 15: continue
        print("| `%s` | %d | %d | %.0f | " % (nm[:60], grid, n, us) + " | ".join("%.4g" % c.get(k, 0) for k in names) + " | %.2f | %.3f |" % (clock, util))

main(sys.argv[1])
