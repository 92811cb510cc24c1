#!/bin/bash
# round-end validation: smoke, the GPU suite, the other BASELINE workloads (short), the 2-rank path on one GPU, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -2
for w in cfg1 cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['value'],1), 'it/s', round(d['ms_per_step'],2), 'ms', d['config']['num_rendered'])"
done
DGM_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 2>gpurun_out/dp2.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dp2 (one GPU shared)', round(d['value'],1), d['n_gpus'], d['config']['parallelism'])" || tail -5 gpurun_out/dp2.err
timeout 900 python bench.py > gpurun_out/bench_end.json 2> gpurun_out/bench_end.err; echo "bench exit $?"
python -c "import json; d=json.load(open('gpurun_out/bench_end.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
