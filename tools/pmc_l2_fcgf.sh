#!/bin/bash
# L2 hit rate of the sparse-convolution kernels (row gathers) with and without the cell-sorted level-0 rows, in the extractor's regime
# (15 rotated copies per backbone pass):   bash tools/pmc_l2_fcgf.sh <outdir>   -> <outdir>/pmc_l2_fcgf.md
# (rocprofv3 counter pass on its own: --kernel-trace + --pmc only)
R=$GRAFT_REPO_ROOT; O=${1:-$R/gpurun_out/r04}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_l2_fcgf.md
for cells in 1 0; do
  rm -rf /tmp/pmc_l2_$cells
  YOHO_FCGF_CELLS=$cells rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/pmc_l2_$cells -- python $R/tools/bench_fcgf.py 300000 2 15 > /tmp/pmc_l2_$cells.log 2>&1
  (echo; echo "== YOHO_FCGF_CELLS=$cells (cell-sorted level-0 rows: $([ $cells = 1 ] && echo on || echo off)); tools/bench_fcgf.py 300000 2 15"; tail -2 /tmp/pmc_l2_$cells.log; echo; python $R/tools/pmc_l2_report.py /tmp/pmc_l2_$cells) >> $O/pmc_l2_fcgf.md 2>&1
done
cat $O/pmc_l2_fcgf.md | cut -c1-220
