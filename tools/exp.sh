#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_quirks.py -m gpu -q 2>&1 | tail -30
