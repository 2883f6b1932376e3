#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conditioning.py -q 2>&1 | tail -3
