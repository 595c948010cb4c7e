# eager per-kernel table (every kernel alone) of the driver-style workload: this tree against a baseline worktree (_basetree);
# (the baseline: `git worktree add -f _basetree <commit> && (cd _basetree && python -c "import __graft_entry__ as g; g.build()")`;
#  _basetree/ is git-ignored, travels to the GPU box with the snapshot, and is removed again with `git worktree remove --force _basetree`)
# NVO_PROF_DETAIL=1 splits the main grid's backward into its launches
F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0 --steps 20 --warmup 5 --late-steps 0"
for t in . _basetree; do
  echo "== tree $t"
  (cd $t && NVO_PROF_DETAIL=1 python bench.py $F 2>&1 >/dev/null | grep "^\[bench\]" | head -${1:-22})
done
