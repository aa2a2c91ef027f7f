python -m pytest tests/test_gpu_kernels.py -q -x -k "des2r or matched" 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -q -x -k "des2r or full" 2>&1 | tail -3
python tools/step_breakdown.py 2>&1 | grep -v amdgpu.ids
python bench.py --no-cpu-baseline --no-dataset --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_repeats'])"
