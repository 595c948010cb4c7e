set -x
python -m pytest tests/test_engine_gpu.py -m gpu -q -k "native_scratch or producer_overflow or deterministic" > gpurun_out/r3_tests7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tests7.log; tail -4 gpurun_out/r3_tests7.log
python tools/ngp_bench.py --profile > gpurun_out/r3_ngp_bench.txt 2>&1; tail -25 gpurun_out/r3_ngp_bench.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_w1b.json 2> gpurun_out/r3_bench_w1b.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_w1b.json
NVO_SHARD_OPT=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_w1_noshard.json 2> gpurun_out/r3_bench_w1_noshard.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_w1_noshard.json
NVO_DIST_CAPTURE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_w1_eager.json 2> gpurun_out/r3_bench_w1_eager.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_w1_eager.json
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3_prof_on -- python3 $ROOT/bench.py --steps 100 --warmup 20 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table > $ROOT/gpurun_out/r3_prof_on.json 2> $ROOT/gpurun_out/r3_prof_on.err
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r3_prof_off -- python3 $ROOT/bench.py --steps 100 --warmup 20 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --no-overlap > $ROOT/gpurun_out/r3_prof_off.json 2> $ROOT/gpurun_out/r3_prof_off.err
cd $ROOT
python tools/rocprof_clean_stats.py gpurun_out/r3_prof_on > gpurun_out/r3_bench_kernel_stats_overlap_on.csv
python tools/rocprof_clean_stats.py gpurun_out/r3_prof_off > gpurun_out/r3_bench_kernel_stats_overlap_off.csv
head -30 gpurun_out/r3_bench_kernel_stats_overlap_on.csv
find gpurun_out/r3_prof_on gpurun_out/r3_prof_off -name "*kernel_trace.csv" -delete; find gpurun_out/r3_prof_on gpurun_out/r3_prof_off -name "*.db" -delete; du -sh gpurun_out/r3_prof_on gpurun_out/r3_prof_off
