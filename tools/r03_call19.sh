#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_fcgf_oracle.py -q 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_fcgf.py -m gpu -q 2>&1 | tail -2
