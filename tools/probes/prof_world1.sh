# rocprofv3 kernel statistics of the process-group step at world size 1 (what N > 1 runs per rank, collectives included)
set -e
mkdir -p gpurun_out
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r5_prof_world1 -- python3 $ROOT/bench.py --gpus 1 --steps 100 --warmup 20 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --render-frames 0 --ngp-steps 0 --pmc-traffic off > $ROOT/gpurun_out/r5_prof_world1.log 2> $ROOT/gpurun_out/r5_prof_world1.err
cd $ROOT
python tools/rocprof_clean_stats.py gpurun_out/r5_prof_world1 > gpurun_out/r5_bench_kernel_stats_world1.csv
head -40 gpurun_out/r5_bench_kernel_stats_world1.csv | cut -c1-150
find gpurun_out/r5_prof_world1 -name "*kernel_trace.csv" -delete
grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r5_prof_world1.log | head -2
