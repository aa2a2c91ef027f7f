#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
export RACE_REPS=40 RACE_SCHEDS=1024x1,2048x1,1024x2,512x2
timeout 300 python tools/debug_chunk_race.py 10000 > $O/race_fixed.log 2>&1
echo "== fixed: $(tail -1 $O/race_fixed.log)"
YOHO_GCONV=fgemm128 timeout 300 python tools/debug_chunk_race.py 10000 > $O/race_fixed_fgemm128.log 2>&1
echo "== fixed fgemm128: $(tail -1 $O/race_fixed_fgemm128.log)"
timeout 400 python tools/sweep_partI_chunk.py 10000 $O/chunk_sweep.json > $O/chunk_sweep.log 2>&1; echo "sweep rc=$?"
tail -12 $O/chunk_sweep.log
timeout 300 python tools/time_partI.py fgemm 10000 > $O/time_partI.log 2>&1; tail -1 $O/time_partI.log
