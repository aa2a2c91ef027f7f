"""HBM traffic per kernel from two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, csv output).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [mode-key]  (prints a markdown table)
FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported; both are KiB."""
import sys, glob, csv, collections, json, re


def load(d, counter):
    acc = collections.OrderedDict()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        disp = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = r.get("Dispatch_Id") or r.get("Dispatch_ID")
            e = disp.setdefault(key, [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""), int(r.get("Grid_Size", 0) or 0), 0.0])
            e[2] += float(r["Counter_Value"])
        for nm, grid, v in disp.values():
            a = acc.setdefault((nm, grid), [0, 0.0])
            a[0] += 1; a[1] += v
    return acc


def main(fd, wd, out, key="fgemm"):
    F, Wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    rows = []
    print("| kernel | grid threads | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (2*F + W) |")
    print("|---|---|---|---|---|---|")
    for k in F:
        n, f = F[k]
        w = Wr.get(k, [n, 0.0])[1] / max(Wr.get(k, [n, 0.0])[0], 1)
        f /= n
        b = (2 * f + w) * 1024
        if b < 30e6:
            continue
        rows.append({"kernel": k[0], "grid": k[1], "dispatches": n, "fetch_kib": f, "write_kib": w, "hbm_bytes_per_launch": b})
        print("| `%s` | %d | %d | %.0f | %.0f | %.0f MB |" % (k[0], k[1], n, f, w, b / 1e6))
    try:
        d = json.load(open(out))
    except Exception:
        d = {"units": "KiB per dispatch as reported by rocprofv3 (FETCH_SIZE/WRITE_SIZE)",
             "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported", "partI_pass_gconv_bytes": {"launches_per_pass": 4}}
    d.setdefault("modes", {})[key] = rows
    if key in ("fgemm", "fgemm256"):
        # one pass = 4 GEMM launches: grids may coincide (layers 1 and 3 share a grid size): weight by dispatch counts
        per = [r for r in rows if "fgemm" in r["kernel"]]
        passes = max(1, min(r["dispatches"] for r in per) if per else 1)
        tot = sum(r["hbm_bytes_per_launch"] * r["dispatches"] for r in per) / passes
        allb = sum(r["hbm_bytes_per_launch"] * r["dispatches"] for r in rows) / passes
        d["partI_pass_gconv_bytes"][key] = tot
        d.setdefault("partI_pass_total_bytes", {})[key] = allb
        print("\nSum over the 4 GEMM launches of one PartI pass: %.2f GB (avg %.2f GB per launch); every kernel of the pass: %.2f GB"
              % (tot / 1e9, tot / 4e9, allb / 1e9))
    import subprocess, os
    try:
        d["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
    except Exception:
        d["commit"] = os.environ.get("YOHO_COMMIT")
    json.dump(d, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
