set -x
NVO_EXPERIMENT_DEEP_HEAD=1 python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --no-kernel-table > gpurun_out/r3_bench_deep.json 2> gpurun_out/r3_bench_deep.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_deep.json
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --no-kernel-table > gpurun_out/r3_bench_p1b.json 2> gpurun_out/r3_bench_p1b.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_p1b.json
python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "pipelined or scratch or graph_replay" > gpurun_out/r3_tests14.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3_tests14.log
