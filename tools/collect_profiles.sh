#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/<tag>/ (copy what should be judged into profiles/).
#   bash tools/collect_profiles.sh r02
set -u
TAG=${1:-run}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
if [ "${ALL_LEGS:-1}" = 1 ]; then python bench.py --all-legs > $OUT/bench_all_legs.json 2> $OUT/bench_all_legs.err; fi
python bench.py --workload cfg4 --steps 40 --warmup 4 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 > $OUT/bench_cfg4_clip8.json 2> $OUT/bench_cfg4_clip8.err
python tools/video_trace.py 96 > $OUT/v_video_runner_time.txt 2> $OUT/v_video.err; python tools/video_trace.py 96 seq >> $OUT/v_video_runner_time.txt 2>> $OUT/v_video.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v -o v -- python tools/video_trace.py 48 seq > $OUT/v_video_one_slot_traced.txt 2>> $OUT/v_video.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d -o d -- python bench.py --no-cpu-baseline --no-kernel-head --no-neck > $OUT/d_bench.json 2> $OUT/d.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e -o e -- python bench.py --streams 1 --frames 24 --no-cpu-baseline --no-kernel-head --no-neck > $OUT/e_bench.json 2> $OUT/e.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q -o q -- python tools/query_time.py 64 > $OUT/q_query64.txt 2> $OUT/q.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k -o k -- python tools/a1_profile.py 16 fp16 > $OUT/k_a1.txt 2> $OUT/k.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq_query -o p -- python tools/query_time.py 24 > /dev/null 2> $OUT/pmc_sq_query.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq_a1 -o p -- python tools/a1_profile.py 16 fp16 > /dev/null 2> $OUT/pmc_sq_a1.err
python tools/pmc_sq_summary.py $OUT/pmc_sq $OUT/pmc_sq_query $OUT/pmc_sq_a1 > $OUT/pmc_sq_summary.txt 2>&1
python tools/r04_kernels.py mixed16 > $OUT/r04_kernels.json 2> $OUT/r04_kernels.err; python tools/r04_kernels.py fp16 >> $OUT/r04_kernels.json 2>> $OUT/r04_kernels.err
python tools/a1_time.py 16 fp16 > $OUT/a1_time.json 2> $OUT/a1_time.err; PH_KHEAD_NO_FALLBACK=1 python tools/a1_time.py 16 fp16 >> $OUT/a1_time.json 2>> $OUT/a1_time.err
if [ "${ALL_LEGS:-1}" = 1 ]; then bash tools/run_train_prof.sh $OUT > $OUT/train_prof.txt 2>&1; fi
python tools/neck_train_time.py 2 > $OUT/neck_train_step.json 2> $OUT/neck_train.err
python tools/gemm32_time.py > $OUT/gemm32_time.txt 2>&1
python tools/pmc_summary.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
ls -R $OUT | head -60
