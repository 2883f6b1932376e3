#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -k "still_materialised" 2>&1 | tail -12 | cut -c1-200
