# the mapping run's early windows (iterations 500 / 2000; few keyframes, clustered samples), EngineConfig.sparse_backward's
# probe armed (auto) and off, one process each
for cfg in "" "NVO_SPARSE_BACKWARD=off"; do
  env $cfg python tools/mapping_loop.py --render-frames 0 --profile-steps 0 --iterations 2400 --keyframes 56 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', round(d['wall_seconds'],3), [(w['first_iteration'], round(w['ms_per_iteration'],4), w['loss_scale']) for w in d['windows']])"
done
