#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "eval_sharded" > $O/pytest_ds.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ds.log
timeout 400 python tools/bench_dataset.py --runs 3 2>/dev/null | tail -1 > $O/bench_dataset_60.json
timeout 600 python tools/bench_dataset.py --preset 3dmatch --runs 2 2>/dev/null | tail -1 > $O/bench_dataset_3dmatch.json
python - <<PY
import json
for f in ("bench_dataset_60","bench_dataset_3dmatch"):
    d=json.loads(open("$O/"+f+".json").read())
    print(f, [(r["page_cache"][:4], r["total_s"], r["pairs_per_s_end_to_end"], r["rank0"]["setup_s (load + H2D + PartI, overlapped)"], r["rank0"]["h2d_plus_partI_s"], r["rank0"]["device_waiting_for_loader_s"], r["rank0"]["ms_per_pair"]) for r in d["runs"]])
PY
