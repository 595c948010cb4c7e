set -x
python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_mapping_gpu.py tests/test_psnr_parity_gpu.py -m gpu -q -x > gpurun_out/r3_tests13.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3_tests13.log
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_p1.json 2> gpurun_out/r3_bench_p1.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_p1.json
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --no-pipeline > gpurun_out/r3_bench_p0.json 2> gpurun_out/r3_bench_p0.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_p0.json
python bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off > gpurun_out/r3_bench_p1_drv.json 2> gpurun_out/r3_bench_p1_drv.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_p1_drv.json
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --optimize-poses > gpurun_out/r3_bench_p1_pose.json 2> gpurun_out/r3_bench_p1_pose.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_p1_pose.json
