#!/bin/bash
# dev aid (round 5): -m gpu tests, then variants alternating with an environment setting
cd "$(dirname "$0")/../.."
tag=${1:-r05}; envs=${2:-}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.txt
tail -4 gpurun_out/${tag}_tests.txt
{
for rep in 1 2; do
for v in variants/*.so; do
for k in "" "$envs"; do
  env $k VDL2GPU_LIB=$PWD/$v python bench.py --no-cpu --no-extra --no-ring --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s %-26s' % ('$v', '$k'), round(d['value']), round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4), d['stats']['repairs'], d['stats']['serial_redos'])
"
done; done; done
} > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
