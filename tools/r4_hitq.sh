set -x
python -m pytest tests/test_tcnn_gpu.py -x -q -k "grid" 2>&1 | tail -5
for q in 0 1; do
  NVO_GRID_HITQ=$q python tools/kernel_bench.py --modes 1 --cases 1 2 --acc-bits 32 2>&1 | grep "grid_bwd"
  NVO_GRID_HITQ=$q python tools/kernel_bench.py --modes 1 --cases 1 2 --acc-bits 32 --dead 0.15 2>&1 | grep "grid_bwd"
done
for q in 0 1; do
  NVO_GRID_HITQ=$q python bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off 2> gpurun_out/r4_hitq${q}.err | grep -o "\"ms_per_step\": [0-9.]*"
  grep "grid_bwd_lds" gpurun_out/r4_hitq${q}.err
done
NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE python tools/kernel_bench.py --modes 1 --cases 1 2 --acc-bits 32 --phase 2>&1 | grep -v "^\[" 
NVO_GRID_HITQ=0 NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE python tools/kernel_bench.py --modes 1 --cases 1 2 --acc-bits 32 --phase 2>&1 | grep -v "^\["
