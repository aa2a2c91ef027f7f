#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
export RACE_REPS=25 RACE_SCHEDS=1024x1,2048x1,1024x2
for v in none ownslice xf1 gemm1 drain "xf1,gemm1"; do
  YOHO_PARTI_DEBUG=$v timeout 200 python tools/debug_chunk_race.py 10000 > $O/race_$v.log 2>&1
  echo "== $v: $(tail -1 $O/race_$v.log)"
done
