set -x
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python __graft_entry__.py --smoke 2>&1 | tail -3
python bench.py 2>&1 | tail -1 > gpurun_out/bench_final.json; cat gpurun_out/bench_final.json | cut -c1-400
python tools/bench_fcgf.py 300000 20 | tail -1
python tools/bench_extract.py 300000 5000 | tail -2
python tools/prof_extract.py 300000 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench_h -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/bench_h.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fcgf_h -- python $GRAFT_REPO_ROOT/tools/bench_fcgf.py 300000 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_bench_h > gpurun_out/bench_h_stats.md 2>&1
python tools/rocpd_stats.py gpurun_out/prof_fcgf_h > gpurun_out/fcgf_h_stats.md 2>&1
tail -1 gpurun_out/bench_h.json | cut -c1-300
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_extract_h -- python $GRAFT_REPO_ROOT/tools/bench_extract.py 300000 5000 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof_extract_h > gpurun_out/extract_h_stats.md 2>&1
python tools/bench_gridnn.py | tail -2
