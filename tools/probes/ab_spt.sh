# A/B of the main forward's samples per thread on the render path (NVO_GRID_FWD_SPT = 1 | 2 | 4)
set -e
mkdir -p gpurun_out
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 10 --psnr off --cpu-baseline off --render-frames 1 --ngp-steps 0 --pmc-traffic off > gpurun_out/ab_$name.log 2>&1
  echo "== $name"; grep -E "grid_fwd\[" gpurun_out/ab_$name.log | cut -c1-110
  python - "$name" <<'P'
import json,sys
for l in open('gpurun_out/ab_%s.log'%sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('  nerfacto', round(d['ms_per_step'],4), 'render', d['render']['frames'][0]['ms_per_frame'])
P
}
run spt2 NVO_GRID_FWD_SPT=2
run spt4 NVO_GRID_FWD_SPT=4
run spt1 NVO_GRID_FWD_SPT=1
