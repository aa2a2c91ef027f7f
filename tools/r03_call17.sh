#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O; cd $R
timeout 600 python tools/stress_determinism.py > $O/stress.log 2>&1; echo "stress rc=$?"; tail -6 $O/stress.log
RACE_REPS=30 RACE_SCHEDS=1024x1,2048x1,1024x2,512x1 timeout 300 python tools/debug_chunk_race.py 10000 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
