# streamed DENSE level of the main grid (level 4): tile-range chunks per bin (NVO_TL_DENSE_CHUNKS; 1 = one accumulate item per bin)
F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0"
for i in 1 2; do
for v in 8 1 2; do
NVO_TL_DENSE_CHUNKS=$v python bench.py $F 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense chunks $v', r['ms_per_step'])"
done; done
