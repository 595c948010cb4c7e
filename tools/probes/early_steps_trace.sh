# which kernels make steps 5..24 of a fresh field slower than steps 205..224: kernel trace of both windows (graph replays)
export TMPDIR=/tmp
ROOT=$(pwd)
F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0 --steps 20 --late-steps 0 --no-kernel-table"
for w in 5 205; do
  rm -rf /tmp/pt_e
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_e -- python3 $ROOT/bench.py $F --warmup $w > /dev/null 2>&1)
  echo "== warmup $w: last 20 steps"; python3 $ROOT/tools/rocprof_clean_stats.py /tmp/pt_e --last 20 2>/dev/null | cut -c1-110 | head -22
done
rm -rf /tmp/pt_e
