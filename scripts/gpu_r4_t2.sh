#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -q -k "long_lists" 2>&1 | tail -15
