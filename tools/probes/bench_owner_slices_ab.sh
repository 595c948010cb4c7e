F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0"
for i in 1 2; do
for v in 24 51; do
NVO_GRID_OWNER_SLICES=$v python bench.py $F 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('owner slices $v', r['ms_per_step'])"
done; done
