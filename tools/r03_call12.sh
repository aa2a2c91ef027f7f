#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "eval_sharded or evaluator" > $O/pytest_ds.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ds.log
timeout 400 python tools/bench_dataset.py --runs 2 2>/dev/null | tail -1 > $O/bench_dataset_w.json
python - <<PY
import json
d=json.loads(open("$O/bench_dataset_w.json").read())
for r in d["runs"]: print(r["page_cache"], r["total_s"], r["pairs_per_s_end_to_end"], r["rank0"])
PY
