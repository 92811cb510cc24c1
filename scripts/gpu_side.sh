#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
b() { python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>&1 | grep '^{' | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["value"], r["ms_per_step"])'; }
echo "side=0: $(DGM_SIDE_STREAM=0 b)"
for m in 1 2 3 1 3; do echo "mode=$m: $(DGM_SIDE_MODE=$m b)"; done
echo "mode=3 prio-1: $(DGM_SIDE_MODE=3 DGM_SIDE_PRIORITY=-1 b)"
echo "mode=1 prio-1: $(DGM_SIDE_MODE=1 DGM_SIDE_PRIORITY=-1 b)"
