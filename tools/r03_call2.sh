#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 300 python tools/debug_chunk_race.py 10000 > $O/race.log 2>&1; echo "race rc=$?"
YOHO_PARTI_DEBUG=serial timeout 300 python tools/debug_chunk_race.py 10000 > $O/race_serial.log 2>&1; echo "race serial rc=$?"
YOHO_PARTI_DEBUG=sideonly timeout 300 python tools/debug_chunk_race.py 10000 > $O/race_sideonly.log 2>&1; echo "race sideonly rc=$?"
timeout 400 python tools/sweep_partI_chunk.py 10000 $O/chunk_sweep.json > $O/chunk_sweep.log 2>&1; echo "sweep rc=$?"
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu.log
grep -c identical $O/race.log; grep differ $O/race.log | head -20
echo serial; grep differ $O/race_serial.log | head -5
echo sideonly; grep differ $O/race_sideonly.log | head -5
tail -12 $O/chunk_sweep.log
