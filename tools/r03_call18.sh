#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "eval_sharded" > $O/pytest_ds.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ds.log
for w in 1 2 3; do
timeout 400 python tools/bench_dataset.py --runs 3 --pair-workers $w 2>$O/ds_w$w.err | tail -1 > $O/bench_dataset_60_w$w.json
python - <<PY
import json
d=json.loads(open("$O/bench_dataset_60_w$w.json").read())
print("workers $w", [(r["page_cache"][:4], r["total_s"], r["pairs_per_s_end_to_end"], r["rank0"]["pairs_s"], r["rank0"]["ms_per_pair"], round(r["registration_recall"],4)) for r in d["runs"]])
PY
done
