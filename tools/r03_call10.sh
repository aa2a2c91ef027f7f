#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/debug_first_region.py plain 2>&1 | tail -5
timeout 200 python tools/debug_first_region.py chunks 2>&1 | tail -5
