#!/bin/bash
# dev aid: durations of the scan-family kernels of one steady-state push for each variants/*.so (rocprofv3 kernel trace)
cd "$(dirname "$0")/../.."
for v in variants/*.so; do
  echo "== $v"
  VDL2GPU_LIB=$PWD/$v scripts/timeline.sh "$@" 2>&1 | awk '$NF ~ /k2a_|k2x_|k2r_|k2s_sort|k2b_|k1_fast/' | head -${TLV_LINES:-14}
done
