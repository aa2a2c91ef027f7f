#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
# bench with the candidate PartI schedules (no CPU baseline / dataset: quick)
for sch in 0 4096x2 5120x2 2048x2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dataset --partI-schedule $sch > $O/bench_sch_$sch.json 2> $O/bench_sch_$sch.err
  python - <<PY
import json
d=json.loads(open("$O/bench_sch_$sch.json").read().strip().splitlines()[-1])
print("$sch", d["ms_per_step"], d["ms_per_step_repeats"], "yohoc", d["yohoc"]["ms_per_step"], "launch", d["roofline_extra"]["launch_ms"], "xf", d["roofline_extra"]["transform_ms"], "pass", d["roofline_extra"]["pass_ms_one_stream"], d["roofline_extra"]["pass_ms_timed_schedule"], d["roofline_extra"]["power"]["timed_steps"])
PY
done
SWEEP_SCHEDS=3072x2,5120x2,4096x2 timeout 300 python tools/sweep_partI_chunk.py 10000 $O/chunk_sweep_b.json > $O/chunk_sweep_b.log 2>&1; tail -5 $O/chunk_sweep_b.log
timeout 400 python tools/bench_dataset.py --runs 2 > $O/bench_dataset.log 2>&1; echo "dataset rc=$?"; tail -c 1800 $O/bench_dataset.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_bench_seq -- python $R/bench.py --no-cpu-baseline --no-dataset --steps 10 --in-flight 1 --repeats 1 > $O/bench_seq.json 2> $O/bench_seq.err
python $R/tools/rocpd_stats.py $O/prof_bench_seq > $O/kernel_trace_bench_seq.md 2>&1
grep -i "cstat\|kabsch_sample\|cprep" $O/kernel_trace_bench_seq.md | head
for sch in 0 1024 4096x2; do
  for cnt in FETCH_SIZE WRITE_SIZE; do
    YOHO_PARTI_CHUNK=$sch PMC_B=10000 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/pmc_${sch}_$cnt -- python $R/tools/pmc_partI.py fgemm > $O/pmc_${sch}_$cnt.log 2>&1
  done
  python $R/tools/pmc_total.py $O/pmc_${sch}_FETCH_SIZE $O/pmc_${sch}_WRITE_SIZE 2 "schedule $sch" | tee -a $O/pmc_schedules.jsonl
done
rm -rf $O/prof_bench_seq $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
