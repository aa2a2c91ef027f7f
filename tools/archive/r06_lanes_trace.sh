#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in 1 2; do
  rm -rf $O/prof_l$L
  YOHO_FCGF_LANES=$L rocprofv3 --kernel-trace -d $O/prof_l$L -- python $R/tools/bench_extract.py 300000 5000 > $O/extract_l$L.log 2>&1
  tail -3 $O/extract_l$L.log
  python $R/tools/trace_lanes.py $O/prof_l$L > $O/lanes_l$L.txt 2>&1
  cat $O/lanes_l$L.txt
  rm -rf $O/prof_l$L
done
