set -x
python -m pytest tests -m gpu -q > gpurun_out/r3_tests9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_tests9.log; tail -6 gpurun_out/r3_tests9.log
for v in 2 1; do NVO_GRID_FWD_SMALL=$v python bench.py --steps 200 --warmup 20 --psnr off --cpu-baseline off > gpurun_out/r3_bench_g_$v.json 2> gpurun_out/r3_bench_g_$v.err; echo "small=$v"; grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/r3_bench_g_$v.json; grep "grid_fwd\|grid_bwd_stream" gpurun_out/r3_bench_g_$v.err; done
python tools/spread_study.py --runs 3 > gpurun_out/r3_spread_study.jsonl 2> gpurun_out/r3_spread_study.err; cat gpurun_out/r3_spread_study.jsonl
