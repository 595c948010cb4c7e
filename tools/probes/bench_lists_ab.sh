F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0"
for i in 1 2; do
for m in auto off on; do
NVO_SPARSE_BACKWARD=$m python bench.py $F 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sparse_backward $m', r['ms_per_step'], 'late', r['late_schedule']['ms_per_step'])"
done; done
