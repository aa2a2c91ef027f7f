#!/bin/bash
# round 3, GPU call 1: chunk sweep, dataset bench, GPU tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python -c "
from yoho_amd.power import PowerMonitor
import time
m=PowerMonitor(0); print('power source', m.source, 'cap', m.cap_w); m.start(); time.sleep(0.3); print(m.stop())
" > $O/power_probe.log 2>&1
timeout 400 python tools/sweep_partI_chunk.py 10000 $O/chunk_sweep.json > $O/chunk_sweep.log 2>&1
echo "sweep rc=$?"
timeout 400 python tools/bench_dataset.py --runs 2 > $O/bench_dataset.log 2>&1
echo "dataset rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -5 $O/pytest_gpu.log
tail -15 $O/chunk_sweep.log
tail -c 1500 $O/bench_dataset.log
cat $O/power_probe.log
