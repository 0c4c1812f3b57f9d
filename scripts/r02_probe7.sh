#!/bin/bash
# k1_fast phase timing (-DK1F_PROF): shader cycles per wavefront-iteration in each phase
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fPIC -shared -I include"
cp vdlm2dec_amd/libvdl2gpu.so /tmp/keep.so
while read -r v; do
  /opt/rocm/bin/hipcc $F -DK1F_PROF $v vdlm2dec_amd/csrc/vdl2gpu.hip -o vdlm2dec_amd/libvdl2gpu.so 2>/dev/null || echo build failed
  echo "variant [$v]"
  python bench.py --no-cpu --no-ring --no-parity --steps 6 --warmup 2 2>&1 >/tmp/b.json | grep -A4 "k1_fast phases"
  tail -1 /tmp/b.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   k1 live', round(d['roofline']['avg_launch_ms'],4), 'alone', round(d['roofline']['alone']['avg_launch_ms'],4))"
done
cp /tmp/keep.so vdlm2dec_amd/libvdl2gpu.so
