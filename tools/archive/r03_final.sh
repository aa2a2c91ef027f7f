#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"], d["roofline"]["frac"], d["roofline"]["conv_total_frac"], d["roofline_extra"]["launch_ms"], d["roofline_extra"]["transform_ms"])
print(d["roofline_extra"]["power"]["timed_steps"])
print(d["dataset"]["runs"][1]["total_s"], d["yohoc"]["ms_per_step"], d["cpu_baseline"]["value"])
print({k:(v["ms"],v["TBps"]) for k,v in d["roofline_extra"]["hbm"].items() if "gft" in k})
PY
wc -l $O/bench_final.json
python - <<PY
import json
d=json.loads(open("$O/bench_final.json").read().strip().splitlines()[-1])
print("selected leg", d.get("yohoo_selected_hypotheses"))
PY
bash tools/collect_profiles.sh ${YOHO_COMMIT:-unknown} > $O/collect.log 2>&1; echo "collect rc=$?"
