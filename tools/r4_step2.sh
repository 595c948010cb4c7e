set -x
python -m pytest tests -x -q -m gpu > gpurun_out/r4_s2_pytest.txt 2>&1; tail -15 gpurun_out/r4_s2_pytest.txt
for v in 0 1; do
  NVO_GRID_L1_FROM_MLP=$v python bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off 2> gpurun_out/r4_s2_l1mlp${v}.err > gpurun_out/r4_s2_l1mlp${v}.json
  grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r4_s2_l1mlp${v}.json
  grep "grid_bwd\|mlp_bwd\|live\|dy_l1" gpurun_out/r4_s2_l1mlp${v}.err
done
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off 2> gpurun_out/r4_s2_200.err | grep -o "\"ms_per_step\": [0-9.]*"
