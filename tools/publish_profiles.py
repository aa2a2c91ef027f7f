#!/usr/bin/env python
"""Copy the summaries tools/collect_profiles.sh left under gpurun_out/<src>/ into profiles/<round>_* (tracked), keeping the
explanatory header of each tracked file (everything before its first table; a new round starts from the previous round's
header) and stamping the commit.

    python tools/publish_profiles.py <commit> [round = r03] [src dir under gpurun_out = r03p]
"""
import json, os, re, shutil, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"
PREV = "r%02d" % (int(RND[1:]) - 1)
SRC, DST = os.path.join(R, "gpurun_out", sys.argv[3] if len(sys.argv) > 3 else "r04p"), os.path.join(R, "profiles")


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
    for name in ("kernel_trace_bench", "kernel_trace_bench_seq", "kernel_trace_bench_timed", "kernel_trace_extract", "pmc_traffic", "pmc_sq"):
        src, dst = os.path.join(SRC, name + ".md"), os.path.join(DST, RND + "_" + name + ".md")
        if not os.path.exists(src):
            print("missing", src)
            continue
        head = header_of(dst) or header_of(os.path.join(DST, PREV + "_" + name + ".md")).replace("Round %d" % int(PREV[1:]), "Round %d" % int(RND[1:]))
        head = head.replace("--no-cpu-baseline --steps 10", "--no-cpu-baseline --no-dataset --repeats 1 --steps 10")
        head = re.sub(r"commit [0-9a-f]{7}", "commit " + commit, head)
        head = head.replace(f"gpurun_out/{PREV}p/", f"gpurun_out/{RND}p/").replace(f"profiles/{PREV}_bench_profiled.json", f"profiles/{RND}_bench_profiled.json")
        body = body_of(src)
        if name in ("kernel_trace_bench", "kernel_trace_bench_seq"):
            # the numbers the header quotes for "this very run": the bench line of the profiled run and the file's own last table
            bj = os.path.join(SRC, "bench.json" if name == "kernel_trace_bench" else "bench_seq.json")
            try:
                rf = json.loads(open(bj).read().strip().splitlines()[-1])["roofline"]
                tab = re.search(r"= ([0-9.]+) / ([0-9.]+) of the fp16 peak", body)
                head = re.sub(r"\(frac [0-9.]+, launch sum [0-9.]+ ms; the table gives [0-9.]+\)",
                              "(frac %.4f, launch sum %.3f ms; the table gives %s)" % (rf["frac"], rf["launch_ms_sum"], tab.group(2) if tab else "n/a"), head)
            except Exception as e:                 # a missing leg must not stop the publication of the others
                print("header numbers of", name, "not refreshed:", e)
        if name == "pmc_traffic":                 # the generated file carries its own header
            body = body[body.index("|"):] if head else body
        open(dst, "w").write(head + body)
        print("wrote", dst)
    shutil.copy(os.path.join(SRC, "pmc_traffic.json"), os.path.join(DST, RND + "_pmc_traffic.json"))
    line = open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(DST, RND + "_bench_profiled.json"), "w").write(line + "\n")
    print(f"wrote {RND}_pmc_traffic.json, {RND}_bench_profiled.json (the bench line of the profiled run; the unprofiled line is {RND}_bench.json)")


if __name__ == "__main__":
    main(sys.argv[1])
