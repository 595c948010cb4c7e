set -x
for b in 256 512 1024; do NVO_POSE_LDS_BLOCK=$b python tools/ngp_bench.py --profile > gpurun_out/r3_ngp_bench_$b.txt 2>&1; grep "pose_bwd\|ms/step" gpurun_out/r3_ngp_bench_$b.txt; done
python -m pytest tests/test_engine_gpu.py -m gpu -q -k "pose" > gpurun_out/r3_tests16.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_tests16.log
