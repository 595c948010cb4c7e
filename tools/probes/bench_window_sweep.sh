# ms/step of 20-step windows at increasing warm-up (the driver times steps 5..24 of a freshly initialised field)
F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0 --no-kernel-table --late-steps 0"
for w in 5 25 65 125 205 405; do
python bench.py $F --steps 20 --warmup $w 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup $w:', round(r['ms_per_step'],4))"
done
