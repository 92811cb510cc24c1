#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python tools/torch_ops_profile.py 10 2>&1 | tail -45
