#!/bin/bash
# round 4, GPU call A: the full -m gpu suite, smoke, the driver's bench command, extractor A/B of the kernel-map construction
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > $O/pytest_gpu_a.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu_a.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_a.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke_a.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"; tail -3 $O/bench_a.err
python - <<PY
import json
d=json.loads(open("$O/bench_a.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"])
r=d["roofline"]; print("roof", r["achieved"], r["frac"], r["frac_pass"], r["frac_step"], r["frac_per_launch"], r["direct_form"]["frac_of_fp16_peak"])
print("sustained", d.get("sustained"))
print("yohoc", {k:v for k,v in d["yohoc"].items() if k!="modes"}); print(d["yohoc"]["modes"]["device_sampling"]["estimator_ms_per_pair"], {k:v for k,v in d["yohoc"]["modes"]["host_parity"].items() if k not in ("contract","note")})
print("fcgf", json.dumps(d.get("fcgf"))[:3000])
print("cpu", json.dumps(d.get("cpu_baseline"))[:1500])
print("dataset", d["dataset"]["runs"][1]["total_s"] if "runs" in d.get("dataset",{}) else d.get("dataset"))
print("range", d["range_guard"]["headline_repeats"], d["range_guard"]["all_legs_repeats"])
PY
for m in full sym; do echo "== YOHO_FCGF_MAPS=$m"; YOHO_FCGF_MAPS=$m timeout 300 python tools/bench_extract.py 300000 5000 2>&1 | tail -3; done
