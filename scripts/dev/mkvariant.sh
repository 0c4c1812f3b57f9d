#!/bin/bash
# dev aid: build the library as variants/<name>.so with extra -D flags: scripts/dev/mkvariant.sh name [-DFOO ...]
cd "$(dirname "$0")/../.."
mkdir -p variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fPIC -shared "$@" vdlm2dec_amd/csrc/vdl2gpu.hip -o variants/$name.so 2>/dev/null && echo built variants/$name.so
