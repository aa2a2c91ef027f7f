#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_per_step_repeats"], d["roofline"]["frac"], d["roofline"]["conv_total_frac"], d["roofline_extra"]["launch_ms"], d["roofline_extra"]["transform_ms"])
print(json.dumps(d.get("dataset"))[:1500])
print(d["cpu_baseline"]["value"], d["yohoc"]["ms_per_step"])
PY
