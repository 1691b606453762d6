#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mtgs_contract.py -x -q -m gpu -k "rigid_object or one_graph" 2>&1 | grep -E "^E  |passed|failed|Error" | head -8
