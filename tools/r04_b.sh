#!/bin/bash
# round 4, GPU call B: head2 (LDS-staged permuted sources), gft16x work stealing (A/B), dataset leg regression check
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x > $O/pytest_gpu_b.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu_b.log
timeout 600 python -m pytest tests/test_gpu_dropin.py -q -x -k "register_pair or chain or partII or scene6 or Lomatch or streamer or dataset_driver_overlapped" > $O/pytest_gpu_b2.log 2>&1; echo "pytest2 rc=$?"; tail -3 $O/pytest_gpu_b2.log
for st in 1 0; do
  for fl in 2 1; do
    YOHO_XF_STEAL=$st timeout 300 python bench.py --steps 20 --warmup 5 --repeats 3 --in-flight $fl --no-cpu-baseline --no-dataset --no-fcgf --no-sustained --no-yohoc > $O/bench_b_s${st}_f${fl}.json 2> /dev/null
    python - <<PY
import json
d=json.loads(open("$O/bench_b_s${st}_f${fl}.json").read().strip().splitlines()[-1])
x=d["roofline_extra"]
print("steal=$st in_flight=$fl ms/step", d["ms_per_step_repeats"]["all_in_order"], "pass", x["pass_ms_one_stream"], "xf", x["transform_ms"], "gemm", x["launch_ms"], "head2?", )
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for st in 1 0; do
  YOHO_XF_STEAL=$st rocprofv3 --kernel-trace --stats -d $O/prof_b_s$st -- python $R/bench.py --no-cpu-baseline --no-dataset --no-fcgf --no-sustained --no-yohoc --repeats 1 --steps 10 > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $O/prof_b_s$st > $O/kernel_trace_b_s$st.md 2>&1
  echo "== steal=$st (streamed)"; grep "gft16x\|head2\|cone1\|fgemm2\|gconv16_kernel<7\|mlp_head\|gft16_kernel<3>" $O/kernel_trace_b_s$st.md | head -12 | cut -c1-160
  rm -rf $O/prof_b_s$st
done
cd $R
echo "== dataset leg standalone"; timeout 300 python tools/bench_dataset.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['runs']: print(r['page_cache'], r['total_s'], r['rank0']['disk_read_s (loader thread)'], r['rank0']['setup_s (load + H2D + PartI, overlapped)'])
"
