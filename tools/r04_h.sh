#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_h.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu_h.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset --no-sustained --no-yohoc > $O/bench_h.json 2> $O/bench_h.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_h.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"])
f=d["fcgf"]; print("fcgf", f.get("ms_per_fragment"), f.get("ms_per_fragment_all")); print(json.dumps(f.get("split_ms"))); print(json.dumps(f.get("phases_ms")))
PY
