# Round evidence, part 2 (run on the GPU box from the repo root; NVO_COMMIT = the commit the tree was built from):
# rocprofv3 kernel statistics of the graph-replayed step with and without stream overlap, of the inference render, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 units applied by
# tools/pmc_traffic.py), the grid forward's L1 counters, the same traffic passes for the occupancy-grid back-end.
set -x
R=${ROUND:-r4}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_prof_on -- python3 $ROOT/bench.py --steps 100 --warmup 20 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --render-frames 0 --ngp-steps 0 > $ROOT/gpurun_out/${R}_prof_on.json 2> $ROOT/gpurun_out/${R}_prof_on.err
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_prof_off -- python3 $ROOT/bench.py --steps 100 --warmup 20 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --render-frames 0 --ngp-steps 0 --no-overlap > $ROOT/gpurun_out/${R}_prof_off.json 2> $ROOT/gpurun_out/${R}_prof_off.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_fetch -- python3 $ROOT/bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --render-frames 0 --ngp-steps 0 > /dev/null 2> $ROOT/gpurun_out/${R}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_write -- python3 $ROOT/bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --render-frames 0 --ngp-steps 0 > /dev/null 2> $ROOT/gpurun_out/${R}_pmc_write.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_fwd_new -- python3 $ROOT/bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --render-frames 0 --ngp-steps 0 > /dev/null 2> $ROOT/gpurun_out/${R}_pmc_fwd_new.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_ngp_fetch -- python3 $ROOT/tools/ngp_bench.py --steps 40 --warmup 20 > /dev/null 2> $ROOT/gpurun_out/${R}_pmc_ngp_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_ngp_write -- python3 $ROOT/tools/ngp_bench.py --steps 40 --warmup 20 > /dev/null 2> $ROOT/gpurun_out/${R}_pmc_ngp_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_prof_render -- python3 $ROOT/bench.py --steps 5 --warmup 2 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --ngp-steps 0 --render-frames 10 > $ROOT/gpurun_out/${R}_prof_render.json 2> $ROOT/gpurun_out/${R}_prof_render.err
cd $ROOT
python tools/rocprof_clean_stats.py gpurun_out/${R}_prof_render > gpurun_out/${R}_render_kernel_stats.csv
python tools/rocprof_clean_stats.py gpurun_out/${R}_prof_on > gpurun_out/${R}_bench_kernel_stats_overlap_on.csv
python tools/rocprof_clean_stats.py gpurun_out/${R}_prof_off > gpurun_out/${R}_bench_kernel_stats_overlap_off.csv
head -30 gpurun_out/${R}_bench_kernel_stats_overlap_on.csv
python tools/pmc_traffic.py gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write gpurun_out/${R}_pmc_fetch_write_per_kernel.json 25
(echo "# grid forward kernels of the default build (k_grid_fwd: main grid, XCD-balanced block map; k_grid_fwd_small: proposal grids)"; python tools/pmc_kernel.py gpurun_out/${R}_pmc_fwd_new k_grid_fwd) > gpurun_out/${R}_pmc_grid_fwd.txt
cat gpurun_out/${R}_pmc_grid_fwd.txt
python tools/pmc_traffic.py gpurun_out/${R}_pmc_ngp_fetch gpurun_out/${R}_pmc_ngp_write gpurun_out/${R}_pmc_ngp_fetch_write_per_kernel.json 60
for d in ${R}_prof_on ${R}_prof_off ${R}_prof_render ${R}_pmc_fetch ${R}_pmc_write ${R}_pmc_fwd_new ${R}_pmc_ngp_fetch ${R}_pmc_ngp_write; do find gpurun_out/$d -name "*kernel_trace.csv" -delete; find gpurun_out/$d -name "*counter_collection.csv" -size +20M -delete; done
du -sh gpurun_out/${R}_p*
