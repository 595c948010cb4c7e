set -x
python -m pytest tests/test_engine_gpu.py -m gpu -q -k "native_scratch or graph_replay or gradscaler" > gpurun_out/r3_tests8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tests8.log; tail -4 gpurun_out/r3_tests8.log
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_f.json 2> gpurun_out/r3_bench_f.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_f.json
NVO_STREAM_OWNER_ACC_BITS=32 python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_f_owner32.json 2> gpurun_out/r3_bench_f_owner32.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_f_owner32.json; grep "grid_bwd_stream" gpurun_out/r3_bench_f.err gpurun_out/r3_bench_f_owner32.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_w1c.json 2> gpurun_out/r3_bench_w1c.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_w1c.json
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3_prof_on -- python3 $ROOT/bench.py --steps 100 --warmup 20 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table > $ROOT/gpurun_out/r3_prof_on.json 2> $ROOT/gpurun_out/r3_prof_on.err
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3_prof_off -- python3 $ROOT/bench.py --steps 100 --warmup 20 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --no-overlap > $ROOT/gpurun_out/r3_prof_off.json 2> $ROOT/gpurun_out/r3_prof_off.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/r3_pmc_fetch -- python3 $ROOT/bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table > /dev/null 2> $ROOT/gpurun_out/r3_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/r3_pmc_write -- python3 $ROOT/bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table > /dev/null 2> $ROOT/gpurun_out/r3_pmc_write.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $ROOT/gpurun_out/r3_pmc_fwd_new -- python3 $ROOT/bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table > /dev/null 2> $ROOT/gpurun_out/r3_pmc_fwd_new.err
export NVO_GRID_FWD_SMALL=0
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $ROOT/gpurun_out/r3_pmc_fwd_old -- python3 $ROOT/bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table > /dev/null 2> $ROOT/gpurun_out/r3_pmc_fwd_old.err
unset NVO_GRID_FWD_SMALL
cd $ROOT
python tools/rocprof_clean_stats.py gpurun_out/r3_prof_on > gpurun_out/r3_bench_kernel_stats_overlap_on.csv
python tools/rocprof_clean_stats.py gpurun_out/r3_prof_off > gpurun_out/r3_bench_kernel_stats_overlap_off.csv
head -40 gpurun_out/r3_bench_kernel_stats_overlap_on.csv
NVO_COMMIT=$(cat gpurun_out/.head 2>/dev/null) python tools/pmc_traffic.py gpurun_out/r3_pmc_fetch gpurun_out/r3_pmc_write gpurun_out/r3_pmc_fetch_write_per_kernel.json 25
(echo "# k_grid_fwd_small (NVO_GRID_FWD_SMALL=1, default)"; python tools/pmc_kernel.py gpurun_out/r3_pmc_fwd_new k_grid_fwd; echo "# k_grid_fwd only (NVO_GRID_FWD_SMALL=0)"; python tools/pmc_kernel.py gpurun_out/r3_pmc_fwd_old k_grid_fwd) > gpurun_out/r3_pmc_grid_fwd.txt
cat gpurun_out/r3_pmc_grid_fwd.txt
for d in r3_prof_on r3_prof_off r3_pmc_fetch r3_pmc_write r3_pmc_fwd_new r3_pmc_fwd_old; do find gpurun_out/$d -name "*kernel_trace.csv" -delete; find gpurun_out/$d -name "*counter_collection.csv" -size +20M -delete; done
du -sh gpurun_out/r3_p*
