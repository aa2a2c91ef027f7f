#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 600 python tools/stress_determinism.py 40 > $O/stress.log 2>&1; echo "stress rc=$?"; tail -6 $O/stress.log
timeout 1500 bash tools/collect_profiles.sh 4315ec1 r04 > $O/collect.log 2>&1; echo "collect rc=$?"; tail -25 $O/collect.log | cut -c1-200
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_d.json 2> $O/bench_d.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_d.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"], "sustained", d["sustained"]["ms_per_step"], d["sustained"]["clock_probe"]["shader_mhz_mean"])
r=d["roofline"]; print("roof", r["achieved"], r["frac"], r["frac_pass"], r["frac_step"], r["frac_per_launch"], r["traffic_source"]["file"])
f=d["fcgf"]; print("fcgf", f.get("ms_per_fragment"), f.get("ms_per_fragment_all"))
print("dataset", [x["total_s"] for x in d["dataset"]["runs"]], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "yohoc", d["yohoc"]["ms_per_step"], "sel", d["yohoo_selected_hypotheses"]["ms_per_step"])
PY
