#!/bin/bash
# The last GPU call of a round: determinism stress, the full -m gpu suite, smoke, the profile collection, the driver's bench command.
#   gpurun --timeout 3000 -- 'bash tools/final_run.sh <commit> <round, e.g. r05>'     then: python tools/publish_profiles.py <commit> <round> <round>p
C=${1:-unknown}; RND=${2:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$RND; mkdir -p $O
cd $R
timeout 600 python tools/stress_determinism.py 40 > $O/stress.log 2>&1; echo "stress rc=$?"; tail -5 $O/stress.log
( time timeout 2400 python -m pytest tests -m gpu -q -s --durations=15 ) > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; grep -a 'CENSUS-SUMMARY\|passed\|failed\|^real' $O/pytest_gpu_final.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 1500 bash tools/collect_profiles.sh $C $RND > $O/collect.log 2>&1; echo "collect rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_final.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], d["ms_per_step_repeats"]["all_in_order"], "sustained", d["sustained"]["ms_per_step"], d["sustained"]["clock_probe"]["shader_mhz_mean"])
r=d["roofline"]; print("roof", r["achieved"], r["frac"], r["frac_at_clock"], r["shader_mhz_mean"], r["power_w"], r["frac_pass"], r["frac_step"], r["frac_per_launch"])
f=d["fcgf"]; print("fcgf", f.get("ms_per_fragment"), f.get("ms_per_fragment_all"), json.dumps(f.get("split_ms")))
f8=d.get("fgemm8") or {}; print("fgemm8+cgemm8", f8.get("ms_per_step"), f8.get("vs_headline_ms_per_step"), f8.get("descriptor_max_abs_diff_vs_default"), f8.get("quaternion_max_abs_diff_vs_default"), f8.get("same_match_list_as_default"), f8.get("same_winner_as_default"), f8.get("partII_cgemm8_only"))
print("traffic", r.get("traffic"), r.get("traffic_note"), (r.get("traffic_detail") or {}).get("over_algorithmic"), "pair_call", d["config"]["pair_call"])
print("yohoc host parity", d["yohoc"]["modes"]["host_parity"]["estimator_ms_per_pair"])
print("dataset", [x["total_s"] for x in d["dataset"]["runs"]], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "yohoc", d["yohoc"]["ms_per_step"], "sel", d["yohoo_selected_hypotheses"]["ms_per_step"])
PY
