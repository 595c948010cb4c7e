# rocprofv3 kernel trace of graph-replayed driver-style steps: this tree against the baseline worktree (_basetree)
# (the baseline: `git worktree add -f _basetree <commit> && (cd _basetree && python -c "import __graft_entry__ as g; g.build()")`;
#  _basetree/ is git-ignored, travels to the GPU box with the snapshot, and is removed again with `git worktree remove --force _basetree`)
export TMPDIR=/tmp
ROOT=$(pwd)
F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0 --steps 100 --warmup 20 --late-steps 0 --no-kernel-table"
for t in . _basetree; do
  rm -rf /tmp/pt_ab
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pt_ab -- python3 $ROOT/$t/bench.py $F > /dev/null 2>&1)
  echo "== tree $t"; python3 $ROOT/tools/rocprof_clean_stats.py /tmp/pt_ab 2>/dev/null | cut -c1-120 | head -${1:-20}
done
rm -rf /tmp/pt_ab
