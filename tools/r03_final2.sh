#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
for i in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final$i.json 2> $O/bench_final$i.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_final$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"], d["roofline"]["frac"], d["roofline"]["conv_total_frac"], d["roofline_extra"]["launch_ms"], d["roofline_extra"]["transform_ms"])
print(d["roofline_extra"]["power"]["timed_steps"], d["roofline_extra"]["power"]["profiled_partI_passes"])
print(d["dataset"]["runs"][1]["total_s"], d["yohoc"]["ms_per_step"], d["cpu_baseline"]["value"])
PY
done
