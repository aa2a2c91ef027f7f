#!/bin/bash
# A/B of the ring depth / register budget of the fine-level sparse convolutions (experiments build, YOHO_SPCONV_VAR), alternating twice.
#   0 = as shipped: 64-channel kernel ring 2 at 4 waves per SIMD, 128-channel ring 2 at 3 waves, 32-channel ring 4 at 4 waves
#   low nibble (64-channel kernel <2>): 1 = ring 4, no budget (3 waves: the kernel up to round 4)   2 = ring 4 at 4 waves (spills)
#                                       3 = ring 6 at 3 waves   4 = ring 2 at 5 waves (28 bytes of scratch)
#   second nibble (32-channel <1>):     0x10 = ring 2 at 5 waves   0x20 = ring 4 at 5 waves (spills)   0x30 = ring 6 at 4 waves
#   0x100: 128-channel kernel <4> with ring 4, no budget (2 waves: up to round 4)
mkdir -p gpurun_out/occ
export YOHO_LIB=exp
for round in 1 2; do
  for v in ${VARS:-0 257 4 16}; do
    echo "== var $v (round $round)"
    YOHO_SPCONV_VAR=$v timeout 300 python tools/bench_fcgf.py 300000 5 15 2>&1 | grep -v "^$\|amdgpu.ids"
  done
done > gpurun_out/occ/ab.log 2>&1
cat gpurun_out/occ/ab.log
