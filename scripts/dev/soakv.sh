#!/bin/bash
# dev aid: scripts/soak.py seeds on each -DVDL2GPU_TESTHOOKS build in variants_t/ (bisecting): scripts/dev/soakv.sh 1009 1019
cd "$(dirname "$0")/../.."
for v in variants_t/*.so; do for seed in "$@"; do echo -n "$v "; VDL2GPU_LIB_TEST=$PWD/$v timeout 120 python scripts/soak.py 1 $seed 2>&1 | grep "^seed" | cut -c1-140; done; done
