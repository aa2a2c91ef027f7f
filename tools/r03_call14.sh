#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -m gpu -q -k "selected or eval_sharded or streamer" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_sel.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset > $O/bench_sel.json 2>$O/bench_sel.err; echo rc=$?
python - <<PY
import json
d=json.loads(open("$O/bench_sel.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"], d["yohoo_selected_hypotheses"], d["yohoc"]["ms_per_step"])
PY
timeout 400 python tools/bench_dataset.py --runs 3 2>/dev/null | tail -1 > $O/bench_dataset_60.json
timeout 600 python tools/bench_dataset.py --preset 3dmatch --runs 2 2>/dev/null | tail -1 > $O/bench_dataset_3dmatch.json
timeout 400 python tools/bench_dataset.py --runs 2 --hypotheses all 2>/dev/null | tail -1 > $O/bench_dataset_60_all.json
timeout 400 python tools/bench_dataset.py --runs 2 --estimator yohoc 2>/dev/null | tail -1 > $O/bench_dataset_60_yohoc.json
python - <<PY
import json
for f in ("bench_dataset_60","bench_dataset_3dmatch","bench_dataset_60_all","bench_dataset_60_yohoc"):
    d=json.loads(open("$O/"+f+".json").read())
    print(f, d["load_weights_once_s"], [(r["page_cache"][:4], r["total_s"], r["pairs_per_s_end_to_end"], r["rank0"]["ms_per_pair"], round(r["registration_recall"],4)) for r in d["runs"]])
PY
