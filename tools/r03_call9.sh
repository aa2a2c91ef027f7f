#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropin.py -m gpu -q -k "transfer or coexist or yoho_extractor or testset_create" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_new.log
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"], "yohoc", d["yohoc"]["ms_per_step"], "launch", d["roofline_extra"]["launch_ms"], "xf", d["roofline_extra"]["transform_ms"], "pass", d["roofline_extra"]["pass_ms_one_stream"])
PY
}
for g in 1 2 4; do
YOHO_PARTI_DEBUG=xfgrid$g timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset > $O/bench_xfgrid$g.json 2>/dev/null; show $O/bench_xfgrid$g.json xfgrid$g
done
timeout 300 python tools/bench_extract.py 300000 5000 > $O/extract.log 2>&1; tail -3 $O/extract.log
