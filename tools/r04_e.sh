#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
(YOHO_LIB=exp timeout 300 python tools/fused_compare.py 10000; timeout 120 tools/_fused_tile_probe 10000 512; timeout 120 tools/_fused_tile_probe 10000 256) 2>&1 | grep -v amdgpu.ids | tee $O/fused_compare.log
