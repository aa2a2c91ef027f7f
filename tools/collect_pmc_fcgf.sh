#!/bin/bash
# SQ counters of the FCGF backbone kernels in the extractor's regime (15 rotated copies per pass) and for one cloud:
#   bash tools/collect_pmc_fcgf.sh      -> gpurun_out/r02/pmc_fcgf.md      (copy into profiles/r02_pmc_fcgf.md)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_fcgf.md
for nb in 15 1; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmc_fcgf_$nb -- python $R/tools/bench_fcgf.py 300000 2 $nb > /dev/null 2>&1
  (echo; echo "== tools/bench_fcgf.py 300000 2 $nb (rotated copies per pass: $nb)"; echo; python $R/tools/pmc_report.py /tmp/pmc_fcgf_$nb | grep -v "hash_\|first_\|block_scan\|coords4\|bbox\|fill_\|parity_\|cell_\|rocclr\|bitmap_fill\|row_normalize\|rotate_sel") >> $O/pmc_fcgf.md 2>&1
done
cat $O/pmc_fcgf.md | cut -c1-200
