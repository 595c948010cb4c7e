NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE python tools/kernel_bench.py --modes 1 --cases 1 2 --acc-bits 32 --phase > gpurun_out/r4_phase_all_live.txt 2>&1
NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE python tools/kernel_bench.py --modes 1 --cases 1 2 --acc-bits 32 --phase --random-x > gpurun_out/r4_phase_random_x.txt 2>&1
grep -v "^\[\|warning\|^ \|note:" gpurun_out/r4_phase_all_live.txt | tail -12
grep -v "^\[\|warning\|^ \|note:" gpurun_out/r4_phase_random_x.txt | tail -12
