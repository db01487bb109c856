#!/bin/bash
# round 6, GPU call E: the fused final stage with the upsample as a second MFMA product (tests + same-box A/B), dist tests, clip phases
set -u
OUT=gpurun_out/r06e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "up2" > $OUT/pytest_up2.log 2>&1; echo "up2 tests rc $?"; tail -25 $OUT/pytest_up2.log
PH_UP2_MFMA=0 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "up2" > $OUT/pytest_up2_old.log 2>&1; echo "up2 tests (window-pass kernel) rc $?"; tail -3 $OUT/pytest_up2_old.log
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['kernels_ms'].items()})" 2>&1 | tail -1; }
for rep in 1 2; do
  python bench.py $Q > $OUT/b_mfma$rep.json 2> $OUT/b_mfma$rep.err; val $OUT/b_mfma$rep.json mfma$rep
  PH_UP2_MFMA=0 python bench.py $Q > $OUT/b_win$rep.json 2> $OUT/b_win$rep.err; val $OUT/b_win$rep.json window$rep
done
python tools/r04_kernels.py mixed16 2> $OUT/k_mfma.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mfma', d['up2_mask_us'], d['up2_depth_us'])"
PH_UP2_MFMA=0 python tools/r04_kernels.py mixed16 2> $OUT/k_win.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window', d['up2_mask_us'], d['up2_depth_us'])"
timeout 900 python -m pytest tests/test_gpu_dist.py -q --durations=6 -k "not two_rank" -s 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/pytest_dist.log; tail -12 $OUT/pytest_dist.log
python tools/clip_phases.py 8 fp16 2> $OUT/clip8.err | tail -1
python tools/clip_phases.py 2 fp16 2> $OUT/clip2.err | tail -1
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x > $OUT/pytest_cfg.log 2>&1; echo "cfg tests rc $?"; tail -4 $OUT/pytest_cfg.log
