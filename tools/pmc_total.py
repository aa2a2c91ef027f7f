"""Fabric-side traffic of a PartI pass from two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; csv), summed over ALL
dispatches (a chunked pass has many small launches, which tools/pmc_traffic.py's per-launch table drops).
usage: pmc_total.py <fetch_dir> <write_dir> <passes> [label]   -> one markdown table row + a JSON line
FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE as reported; both are KiB."""
import sys, glob, csv, collections, json, re


def load(d, counter):
    acc = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("yoho::", "")
            grp = "GEMM" if "fgemm" in nm else ("transform" if "gft16" in nm else ("head/tail" if ("head16" in nm or "finalize" in nm) else "other"))
            acc[grp] += float(r["Counter_Value"])
    return acc


fd, wd, passes = sys.argv[1], sys.argv[2], float(sys.argv[3])
label = sys.argv[4] if len(sys.argv) > 4 else ""
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
out = {"label": label}
tot = 0.0
for g in ("GEMM", "transform", "head/tail"):
    b = (2 * F.get(g, 0.0) + W.get(g, 0.0)) * 1024 / passes
    out[g + "_GB"] = round(b / 1e9, 3)
    tot += b
out["total_GB"] = round(tot / 1e9, 3)
out["fetch_GB"] = round(sum(2 * F[g] for g in ("GEMM", "transform", "head/tail")) * 1024 / passes / 1e9, 3)
out["write_GB"] = round(sum(W[g] for g in ("GEMM", "transform", "head/tail")) * 1024 / passes / 1e9, 3)
print(json.dumps(out))
