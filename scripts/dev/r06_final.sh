#!/bin/bash
# round 6, the committed tree on one box: GPU tests, profiles (headline + legs), stage Gantt, BER curve, both soaks
#   R_HEAD=<commit> scripts/dev/r06_final.sh        (copy gpurun_out/profiles/* to profiles/ afterwards)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/profiles
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r06_final_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06_final_tests.txt; tail -3 gpurun_out/r06_final_tests.txt
R=r06 WITH_BER=1 bash scripts/make_profiles.sh 2>&1 | tail -30
R=r06 bash scripts/make_profiles_configs.sh c3 c4 busy15 busy30 2>&1 | tail -8
cp vdlm2dec_amd/kernel_resources.txt gpurun_out/profiles/r06_kernel_resources.txt
VDL2GPU_STAGE_DUMP=1 timeout 300 python bench.py --no-cpu --no-extra --no-ring --no-parity --steps 32 --warmup 5 2>/tmp/gantt_err.txt >/dev/null
python scripts/dev/stage_gantt.py /tmp/gantt_err.txt > gpurun_out/profiles/r06_stage_gantt.txt 2>&1; tail -5 gpurun_out/profiles/r06_stage_gantt.txt
timeout 500 python bench.py --gpus 2 --backend gloo --share-gpu --steps 8 --warmup 3 2>/dev/null | tail -1 > gpurun_out/profiles/r06_bench_line_2ranks_gloo_shared_gpu.json
( timeout 700 python scripts/soak.py 560 8000; ) > gpurun_out/profiles/r06_soak.txt 2>&1
( timeout 700 python scripts/soak_pipeline.py 560 8000; ) >> gpurun_out/profiles/r06_soak.txt 2>&1
grep "^soak" gpurun_out/profiles/r06_soak.txt
