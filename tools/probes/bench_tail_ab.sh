# same-box A/B of EngineConfig.overlap_optimizer_tail on the driver-style workload (NVO_OVERLAP_TAIL=0: the sequential tail)
F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0"
for i in 1 2 3; do
python bench.py $F 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail beside accumulate', r['ms_per_step'])"
NVO_OVERLAP_TAIL=0 python bench.py $F 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail behind accumulate', r['ms_per_step'])"
done
