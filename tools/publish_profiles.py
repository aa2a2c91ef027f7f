#!/usr/bin/env python
"""Copy the summaries tools/collect_profiles.sh left under gpurun_out/r02/ into profiles/r02_* (tracked), keeping the
explanatory header of each tracked file (everything before its first table) and stamping the commit.

    python tools/publish_profiles.py <commit>
"""
import json, os, re, shutil, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(R, "gpurun_out", "r02"), os.path.join(R, "profiles")


def header_of(path):
    if not os.path.exists(path):
        return ""
    out = []
    for line in open(path):
        if line.startswith("|") or line.startswith("== "):
            break
        out.append(line)
    return "".join(out)


def body_of(path, drop_prefixes=("+ ", "fatal:")):
    return "".join(l for l in open(path) if not l.startswith(drop_prefixes))


def main(commit):
    for name in ("kernel_trace_bench", "kernel_trace_bench_seq", "kernel_trace_extract", "pmc_traffic", "pmc_sq"):
        src, dst = os.path.join(SRC, name + ".md"), os.path.join(DST, "r02_" + name + ".md")
        if not os.path.exists(src):
            print("missing", src)
            continue
        head = re.sub(r"commit [0-9a-f]{7}", "commit " + commit, header_of(dst))
        body = body_of(src)
        if name == "pmc_traffic":                 # the generated file carries its own header
            body = body[body.index("|"):] if head else body
        open(dst, "w").write(head + body)
        print("wrote", dst)
    shutil.copy(os.path.join(SRC, "pmc_traffic.json"), os.path.join(DST, "r02_pmc_traffic.json"))
    line = open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(DST, "r02_bench_profiled.json"), "w").write(line + "\n")
    print("wrote r02_pmc_traffic.json, r02_bench_profiled.json (the bench line of the profiled run; the unprofiled line is r02_bench.json)")


if __name__ == "__main__":
    main(sys.argv[1])
