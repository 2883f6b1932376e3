#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py -x -q -k "cnmf" 2>&1 | tail -8 | cut -c1-220
