set -x
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_n1_driver_style.json 2> gpurun_out/r3_bench_n1_driver_style.err; echo "rc=$?"; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_n1_driver_style.json; grep -o "\"render_psnr\": {.*" gpurun_out/r3_bench_n1_driver_style.json | cut -c1-900
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_n1.json
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --optimize-poses > gpurun_out/r3_bench_n1_optimize_poses.json 2> gpurun_out/r3_bench_n1_optimize_poses.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_n1_optimize_poses.json
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --workload replica360 > gpurun_out/r3_bench_n1_replica360.json 2> gpurun_out/r3_bench_n1_replica360.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_n1_replica360.json
python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off --workload scannet > gpurun_out/r3_bench_n1_scannet.json 2> gpurun_out/r3_bench_n1_scannet.err; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_n1_scannet.json
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/r3_pmc_ngp_fetch -- python3 $ROOT/tools/ngp_bench.py --steps 40 --warmup 20 > /dev/null 2> $ROOT/gpurun_out/r3_pmc_ngp_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/r3_pmc_ngp_write -- python3 $ROOT/tools/ngp_bench.py --steps 40 --warmup 20 > /dev/null 2> $ROOT/gpurun_out/r3_pmc_ngp_write.err
cd $ROOT
python tools/pmc_traffic.py gpurun_out/r3_pmc_ngp_fetch gpurun_out/r3_pmc_ngp_write gpurun_out/r3_pmc_ngp_fetch_write_per_kernel.json 60
for d in r3_pmc_ngp_fetch r3_pmc_ngp_write; do find gpurun_out/$d -name "*kernel_trace.csv" -delete; done
