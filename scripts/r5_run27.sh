#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -k "cnmfsc" 2>&1 | tail -6
