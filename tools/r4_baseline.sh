# Round-4 starting point on one MI355X: the step in the reference's loss-scale regime (GradScaler, 65536 dynamic) next to
# the static-128 regime, the all-live slice-owner launches of the proposal grids alone, and their phase clocks.
set -x
R=r4
python bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --dynamic-loss-scale > gpurun_out/${R}_base_dyn_driver.json 2> gpurun_out/${R}_base_dyn_driver.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/${R}_base_dyn_driver.json
python bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off > gpurun_out/${R}_base_static_driver.json 2> gpurun_out/${R}_base_static_driver.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/${R}_base_static_driver.json
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --dynamic-loss-scale > gpurun_out/${R}_base_dyn.json 2> gpurun_out/${R}_base_dyn.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/${R}_base_dyn.json
python tools/kernel_bench.py --modes 1 --cases 1 2 --acc-bits 32 > gpurun_out/${R}_base_kernel_bench.txt 2>&1; cat gpurun_out/${R}_base_kernel_bench.txt | tail -8
NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE python tools/grid_phase.py --dynamic-loss-scale > gpurun_out/${R}_base_phase_dyn.txt 2>&1; tail -30 gpurun_out/${R}_base_phase_dyn.txt
