#!/bin/bash
# Round 6: do non-temporal coefficient stores in fgemm3 and non-temporal LDS-DMA loads in gft16x cut the fabric-side
# traffic of a PartI pass?  Experiments build (YOHO_EXPERIMENTS=1 python -m yoho_amd.build), A/B inside one gpurun call:
#   bash tools/nt_traffic_ab.sh  ->  gpurun_out/r6nt/{base,stnt,stnt_ldnt,sc1,sc1_ldnt}_{traffic.md,time.txt}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6nt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export YOHO_LIB=exp
run() {   # tag, YOHO_FGEMM_DEBUG, YOHO_PARTI_DEBUG
  tag=$1
  for cnt in FETCH_SIZE WRITE_SIZE; do
    YOHO_FGEMM_DEBUG=$2 YOHO_PARTI_DEBUG=$3 PMC_B=10000 rocprofv3 --kernel-trace --pmc $cnt --output-format csv -d $O/${tag}_$cnt -- python $R/tools/pmc_partI.py fgemm > $O/${tag}_$cnt.log 2>&1
  done
  (cd $R && python tools/pmc_traffic.py $O/${tag}_FETCH_SIZE $O/${tag}_WRITE_SIZE $O/${tag}_traffic.json fgemm > $O/${tag}_traffic.md 2>&1)
  for rep in 1 2 3; do YOHO_FGEMM_DEBUG=$2 YOHO_PARTI_DEBUG=$3 python $R/tools/time_partI.py fgemm 10000 >> $O/${tag}_time.txt 2>&1; done
  rm -rf $O/${tag}_FETCH_SIZE $O/${tag}_WRITE_SIZE
  echo "== $tag"; tail -2 $O/${tag}_traffic.md; tail -3 $O/${tag}_time.txt
}
run base none none
run stnt stnt none
run stnt_ldnt stnt ldnt
run sc1 sc1 none
run sc1_ldnt sc1 ldnt
run ldnt none ldnt
