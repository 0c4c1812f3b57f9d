#!/bin/bash
# dev aid: rebuild libvdl2gpu.so with extra -D flags on the GPU box and bench each variant
#   scripts/variants.sh "-DK2A_THREADS=512" "-DK2A_TS=512" ...
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fPIC -shared -I include"
cp vdlm2dec_amd/libvdl2gpu.so /tmp/libvdl2gpu.keep
for v in "$@"; do
  /opt/rocm/bin/hipcc $F $v vdlm2dec_amd/csrc/vdl2gpu.hip -o vdlm2dec_amd/libvdl2gpu.so 2>/dev/null || { echo "variant $v: build failed"; continue; }
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu 2>&1 | grep "^{" > /tmp/kv.json
  python -c "import json; d=json.load(open('/tmp/kv.json')); print('variant $v', round(d['value']), d['ms_per_step'], d['kernels_ms'], d['parity']['equal'])"
done
cp /tmp/libvdl2gpu.keep vdlm2dec_amd/libvdl2gpu.so
