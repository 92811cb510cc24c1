#!/bin/bash
# the BASELINE configs: one stream, two streams, and what the calibration picks (informational)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
b() { python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline --workload $1 2>&1 | grep '^{' | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(round(r["value"],1), round(r["ms_per_step"],3), "streams", r["streams"], r.get("stream_calibration"))'; }
for w in cfg1 cfg2 cfg3 cfg4 cfg5; do
  echo "$w auto: $(b $w)"
done
timeout 900 python -m pytest tests/test_trainer_dp_gpu.py tests/test_rccl.py -m gpu -q 2>&1 | tail -2
