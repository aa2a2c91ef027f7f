#!/bin/bash
# Round-end profile collection on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <commit> [round = r06]     -> gpurun_out/<round>p/*   (then: python tools/publish_profiles.py <commit> <round> <round>p)
# rocprofv3 counter passes are separate runs with --kernel-trace only (no --stats / sys-trace next to --pmc).
set -x
export YOHO_COMMIT=${1:-unknown}
RND=${2:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RND}p
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-dataset --no-fcgf --no-sustained --no-traffic"
# 1. kernel trace of the bench command (pairs strictly one after the other, so that per-kernel durations are not inflated by overlap)
rocprofv3 --kernel-trace --stats -d $O/prof_bench_seq -- python $R/bench.py $LEGS --repeats 1 --steps 10 --in-flight 1 > $O/bench_seq.json 2> $O/bench_seq.err
python $R/tools/rocpd_stats.py $O/prof_bench_seq > $O/kernel_trace_bench_seq.md 2>&1
# 2. the same with the default two pairs in flight
rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py $LEGS --repeats 1 --steps 10 > $O/bench.json 2> $O/bench.err
python $R/tools/rocpd_stats.py $O/prof_bench > $O/kernel_trace_bench.md 2>&1
# 2b. the timed steps alone (warm-up + the headline's timed regions, nothing else): what a step launches - no tensor-library kernel
rocprofv3 --kernel-trace --stats -d $O/prof_bench_timed -- python $R/bench.py --timed-only --repeats 1 --steps 10 > $O/bench_timed.json 2> $O/bench_timed.err
python $R/tools/rocpd_stats.py $O/prof_bench_timed > $O/kernel_trace_bench_timed.md 2>&1
# 3. HBM traffic of one PartI pass over 10000 keypoints: FETCH_SIZE and WRITE_SIZE in separate passes
for cnt in FETCH_SIZE WRITE_SIZE; do
  PMC_B=10000 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/pmc_$cnt -- python $R/tools/pmc_partI.py fgemm > $O/pmc_$cnt.log 2>&1
done
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json fgemm > $O/pmc_traffic.md 2>&1
cd /tmp
# 4. SQ counters: matrix-pipe utilisation and effective clock of the PartI kernels
PMC_B=10000 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq_plain -- python $R/tools/pmc_partI.py fgemm > $O/pmc_sq_plain.log 2>&1
(echo "== default build"; python $R/tools/pmc_report.py $O/pmc_sq_plain) >> $O/pmc_sq.md 2>&1
# 5. the one-cloud extractor (FCGF backbone x60 + feature transfer + PartI)
rocprofv3 --kernel-trace --stats -d $O/prof_extract -- python $R/tools/bench_extract.py 300000 5000 > $O/extract.log 2>&1
python $R/tools/rocpd_stats.py $O/prof_extract > $O/kernel_trace_extract.md 2>&1
# 6. L2 hit rate of the sparse-convolution row gathers with / without cell-sorted level-0 rows
bash $R/tools/pmc_l2_fcgf.sh $O > $O/pmc_l2.log 2>&1
# 7. (round 4 only: the fused all-60-coefficient tile probe, tools/archive/fused_tile_probe.hip - run when its binary is present)
[ -x $R/tools/_fused_tile_probe ] && $R/tools/_fused_tile_probe 10000 512 > $O/fused_tile_probe.log 2>&1
cd $R
# 8. the dataset-scale rows (profiles/rNN_dataset.md)
python tools/dataset_profile.py $O/dataset.md > $O/dataset.log 2>&1
# keep only the summaries (the raw databases / csv stay on the box)
rm -rf $O/prof_bench_seq $O/prof_bench $O/prof_bench_timed $O/prof_extract $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq_plain
ls -la $O
tail -c 400 $O/bench.json; echo; tail -3 $O/extract.log; head -12 $O/pmc_traffic.md
