#!/bin/bash
# dev aid (round 5): -m gpu tests, then configs legs under environment settings alternating
cd "$(dirname "$0")/../.."
tag=${1:-r05}; legs=${2:-c3}; shift; shift
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.txt
tail -4 gpurun_out/${tag}_tests.txt
fi
{
for rep in 1 2; do
for k in "$@"; do
  echo "== $k"; env $k python scripts/dev/legs.py $legs 2>/dev/null
done; done
} > gpurun_out/${tag}_legs.txt 2>&1
cat gpurun_out/${tag}_legs.txt
