#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
df -h /tmp | tail -1
timeout 600 python tools/bench_dataset.py --preset 3dmatch --runs 2 > $O/bench_dataset_3dmatch.log 2>&1; echo "3dmatch rc=$?"; tail -c 2500 $O/bench_dataset_3dmatch.log
bash tools/collect_profiles.sh $(cat $R/.commit_id 2>/dev/null || echo unknown) > $O/collect.log 2>&1; echo "collect rc=$?"; tail -30 $O/collect.log
