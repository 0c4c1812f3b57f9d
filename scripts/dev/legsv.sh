#!/bin/bash
# dev aid: bench.py's `configs` legs (scripts/dev/legs.py) for each variants/*.so, alternating: scripts/dev/legsv.sh 2 busy30 c3
cd "$(dirname "$0")/../.."
reps=${1:-2}; shift
for rep in $(seq $reps); do for v in variants/*.so; do echo -n "$v "; VDL2GPU_LIB=$PWD/$v python scripts/dev/legs.py "$@" 2>/dev/null | cut -c1-160 | tr '\n' ' '; echo; done; done
