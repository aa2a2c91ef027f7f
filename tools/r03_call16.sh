#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout 400 python tools/bench_dataset.py --runs 2 --hypotheses all 2>/dev/null | tail -1 > $O/bench_dataset_60_all.json
timeout 400 python tools/bench_dataset.py --runs 2 --estimator yohoc 2>/dev/null | tail -1 > $O/bench_dataset_60_yohoc.json
timeout 600 python tools/bench_dataset.py --preset 3dmatch --runs 2 --estimator yohoc 2>/dev/null | tail -1 > $O/bench_dataset_3dmatch_yohoc.json
python - <<PY
import json
for f in ("bench_dataset_60_all","bench_dataset_60_yohoc","bench_dataset_3dmatch_yohoc"):
    d=json.loads(open("$O/"+f+".json").read())
    print(f, [(r["page_cache"][:4], r["total_s"], r["pairs_per_s_end_to_end"], r["rank0"]["ms_per_pair"]) for r in d["runs"]])
PY
