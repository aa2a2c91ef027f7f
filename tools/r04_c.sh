#!/bin/bash
# round 4, GPU call C: batched feature transfer, fused-tile probe, L2 hit rate of the sparse convolutions, full bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fcgf.py tests/test_gpu_dropin.py -q -x -k "fcgf or extractor or testset or backbone or grid or transfer or gather" > $O/pytest_gpu_c.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu_c.log
echo "== fused tile probe"; timeout 120 tools/_fused_tile_probe 10000 512 2>&1 | tee $O/fused_tile_probe.log
for m in staged batched; do echo "== YOHO_TRANSFER=$m"; YOHO_TRANSFER=$m timeout 300 python tools/bench_extract.py 300000 5000 2>&1 | tail -2; done
bash tools/pmc_l2_fcgf.sh $O > $O/pmc_l2.log 2>&1; tail -40 $O/pmc_l2_fcgf.md | cut -c1-200
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c.json 2> $O/bench_c.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_c.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"], "sustained", d["sustained"]["ms_per_step"], d["sustained"]["clock_probe"])
r=d["roofline"]; print("roof", r["achieved"], r["frac"], r["frac_pass"], r["frac_step"], r["frac_per_launch"])
f=d["fcgf"]; print("fcgf", f.get("ms_per_fragment"), f.get("ms_per_fragment_all"), json.dumps(f.get("split_ms")))
print("dataset", [x["total_s"] for x in d["dataset"]["runs"]], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
x=d["roofline_extra"]; print("launch", x["launch_ms"], "xf", x["transform_ms"], "pass", x["pass_ms_one_stream"])
PY
