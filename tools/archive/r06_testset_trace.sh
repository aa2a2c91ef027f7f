#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in 2 1; do
  rm -rf $O/prof_t$L
  BENCH_TESTSET_LANES=$L,$L rocprofv3 --kernel-trace -d $O/prof_t$L -- python $R/tools/bench_testset.py 4 > $O/testset_l$L.log 2>&1
  tail -3 $O/testset_l$L.log
  python $R/tools/trace_lanes.py $O/prof_t$L 500 200 > $O/lanes_t$L.txt 2>&1
  cat $O/lanes_t$L.txt
  rm -rf $O/prof_t$L
done
