python -m pytest tests/test_engine_gpu.py -x -q -k "render or eval" > gpurun_out/r4_render_pytest.txt 2>&1; tail -3 gpurun_out/r4_render_pytest.txt
python -m pytest tests/test_mapping_gpu.py tests/test_evaluation_gpu.py tests/test_dataset_golden_gpu.py tests/test_ngp_ingest_golden.py -x -q > gpurun_out/r4_render_pytest2.txt 2>&1; tail -3 gpurun_out/r4_render_pytest2.txt
python bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 > gpurun_out/r4_render_bench.json 2> gpurun_out/r4_render_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_render_bench.json'))
print(d['ms_per_step'])
for f in d['render']['frames']: print(f)
PY
grep "render\|^\[bench\]   " gpurun_out/r4_render_bench.err | tail -25
