#!/bin/bash
# dev aid (round 5): -m gpu tests, per-variant kernel stats, the stage Gantt of the current library
cd "$(dirname "$0")/../.."
tag=${1:-r05}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.txt
tail -4 gpurun_out/${tag}_tests.txt
timeout 600 scripts/dev/kstatsv.sh > gpurun_out/${tag}_ks.txt 2>&1
cat gpurun_out/${tag}_ks.txt
timeout 300 scripts/dev/gantt_env.sh > gpurun_out/${tag}_gantt.txt 2>&1
cat gpurun_out/${tag}_gantt.txt
