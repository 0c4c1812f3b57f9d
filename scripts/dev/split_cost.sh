#!/bin/bash
# dev aid: what does a part cost?  The headline push whole, in two and in four parts (VDL2GPU_SPLIT_SAMPLES), with the calling thread's profile
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for k in "" "VDL2GPU_SPLIT_SAMPLES=33600000" "VDL2GPU_SPLIT_SAMPLES=16800000"; do
  env VDL2GPU_LIB=$PWD/vdlm2dec_amd/libvdl2gpu_test.so VDL2GPU_HOST_PROF=1 $k python bench.py --no-cpu --no-extra --no-ring --no-parity --steps 32 --warmup 5 2>/tmp/split_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$k', round(d['value']), round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4), d['host_ms_per_step']['in_push'], d['host_ms_per_step']['in_poll_ready'])
"
  grep "host profile" /tmp/split_err.txt
done; done
