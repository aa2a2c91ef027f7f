#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
show() { python - <<PY
import json
d=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"])
PY
}
YOHO_BENCH_SMU=1 YOHO_BENCH_PROBE_US=20 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset --no-yohoc > $O/b_a.json 2>/dev/null; show $O/b_a.json smu+probe
YOHO_BENCH_SMU=0 YOHO_BENCH_PROBE_US=20 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset --no-yohoc > $O/b_b.json 2>/dev/null; show $O/b_b.json probe_only
YOHO_BENCH_SMU=1 YOHO_BENCH_PROBE_US=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset --no-yohoc > $O/b_c.json 2>/dev/null; show $O/b_c.json smu_only
YOHO_BENCH_SMU=0 YOHO_BENCH_PROBE_US=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset --no-yohoc > $O/b_d.json 2>/dev/null; show $O/b_d.json none
