#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
p=d["roofline_extra"]["power"]["timed_steps"]
print("$2", d["ms_per_step"], d["ms_per_step_repeats"]["min"], d["ms_per_step_repeats"]["max"], "yohoc", d["yohoc"]["ms_per_step"], "launch", d["roofline_extra"]["launch_ms"], "xf", d["roofline_extra"]["transform_ms"], "pass", d["roofline_extra"]["pass_ms_one_stream"], "sclk", p["sclk_mhz_mean"], "W", p["power_w_mean"], "probe", p["clock_probe"])
PY
}
YOHO_BENCH_PROBE_US=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset > $O/bench_noprobe.json 2>/dev/null; show $O/bench_noprobe.json noprobe
YOHO_BENCH_PROBE_US=20 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset > $O/bench_probe20.json 2>/dev/null; show $O/bench_probe20.json probe20
YOHO_BENCH_PROBE_US=1000 YOHO_BENCH_PROBES_PER_STEP=6 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset > $O/bench_probe1000.json 2>/dev/null; show $O/bench_probe1000.json probe1000
YOHO_BENCH_PROBE_US=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset --repeats 1 > $O/bench_noprobe_r1.json 2>/dev/null; show $O/bench_noprobe_r1.json noprobe_rep1
