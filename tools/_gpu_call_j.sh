set -x
python -m pytest tests/test_engine_gpu.py tests/test_ngp_gpu.py tests/test_fullsize_gpu.py -m gpu -q > gpurun_out/r3_tests15.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3_tests15.log
python tools/ngp_bench.py --profile > gpurun_out/r3_ngp_bench.txt 2>&1; grep "pose_bwd\|ema_update\|ms/step" gpurun_out/r3_ngp_bench.txt
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --optimize-poses > gpurun_out/r3_bench_pose.json 2> gpurun_out/r3_bench_pose.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_pose.json; grep "pose_bwd" gpurun_out/r3_bench_pose.err
