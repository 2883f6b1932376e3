#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_golden.py tests/test_gpu_pins.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
