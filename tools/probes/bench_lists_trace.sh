# per-kernel durations of graph-replayed driver-style steps with the round-6 lists armed / not armed (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0 --steps 100 --warmup 20 --late-steps 0 --no-kernel-table"
for m in on off; do
  rm -rf /tmp/pl_$m
  if [ $m = off ]; then export NVO_MLP_SKIP_DEAD=0 NVO_GRID_LIVE_ROWS=0; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_$m -- python3 $ROOT/bench.py $F > /dev/null 2>&1
  echo "== lists $m"; python3 $ROOT/tools/rocprof_clean_stats.py /tmp/pl_$m 2>/dev/null | cut -c1-110 | head -24
done
