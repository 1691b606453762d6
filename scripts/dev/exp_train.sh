#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mtgs_contract.py -x -q -m gpu -k "touch_first" 2>&1 | grep -E "^E  |passed|failed" | head -8
