F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0"
for i in 1 2; do
python bench.py $F 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lists on ', r['ms_per_step']); json.dump(r.get('kernel_table'), open('gpurun_out/kt_on.json','w'))"
NVO_MLP_SKIP_DEAD=0 NVO_GRID_LIVE_ROWS=0 python bench.py $F 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lists off', r['ms_per_step']); json.dump(r.get('kernel_table'), open('gpurun_out/kt_off.json','w'))"
done
