#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "world2 or eval_sharded" 2>&1 | tail -15
