# Round evidence, part 3 (run on the GPU box from the repo root; NVO_COMMIT = the commit the tree was built from): the
# occupancy-grid back-end after the second half of round 4 -- profiled step + 1200x680 inference, rocprofv3 kernel
# statistics of the graph-replayed steps, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), the driver-style line.
set -x
R=${ROUND:-r4}
ROOT=$GRAFT_REPO_ROOT
python tools/ngp_bench.py --steps 300 --profile --render-frames 3 > gpurun_out/${R}_ngp_bench.txt 2>&1; grep -E "ms/step|render 1200" gpurun_out/${R}_ngp_bench.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_n1_driver_style.json 2> gpurun_out/${R}_bench_n1_driver_style.err; echo "rc=$?"; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/${R}_bench_n1_driver_style.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_prof_ngp -- python3 $ROOT/tools/ngp_bench.py --steps 200 > $ROOT/gpurun_out/${R}_prof_ngp.log 2> $ROOT/gpurun_out/${R}_prof_ngp.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_ngp_fetch -- python3 $ROOT/tools/ngp_bench.py --steps 40 --warmup 20 > /dev/null 2> $ROOT/gpurun_out/${R}_pmc_ngp_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_ngp_write -- python3 $ROOT/tools/ngp_bench.py --steps 40 --warmup 20 > /dev/null 2> $ROOT/gpurun_out/${R}_pmc_ngp_write.err
cd $ROOT
python tools/rocprof_clean_stats.py gpurun_out/${R}_prof_ngp --head k_rays_given --skip-first 320 > gpurun_out/${R}_ngp_kernel_stats.csv
head -24 gpurun_out/${R}_ngp_kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/${R}_pmc_ngp_fetch gpurun_out/${R}_pmc_ngp_write gpurun_out/${R}_pmc_ngp_fetch_write_per_kernel.json 60
for d in ${R}_prof_ngp ${R}_pmc_ngp_fetch ${R}_pmc_ngp_write; do find gpurun_out/$d -name "*kernel_trace.csv" -delete; find gpurun_out/$d -name "*counter_collection.csv" -size +20M -delete; done
