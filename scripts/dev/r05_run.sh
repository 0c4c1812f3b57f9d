#!/bin/bash
# dev aid: one gpurun call of round 5 -- the -m gpu tests, then the library variants variants/*.so alternating on the same box
# usage: scripts/dev/r05_run.sh tag [reps] [pytest args...]
cd "$(dirname "$0")/../.."
tag=${1:-r05}; reps=${2:-2}; shift; shift
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/${tag}_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.txt
tail -5 gpurun_out/${tag}_tests.txt
timeout 900 scripts/dev/abv.sh $reps > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
