# bottom line on one box: this tree against the baseline worktree (_basetree), driver-style workload, 200 timed steps
# (the baseline: `git worktree add -f _basetree <commit> && (cd _basetree && python -c "import __graft_entry__ as g; g.build()")`;
#  _basetree/ is git-ignored, travels to the GPU box with the snapshot, and is removed again with `git worktree remove --force _basetree`)
F="--mapping-loop off --pmc-traffic off --no-cpu-baseline --psnr off --ngp-steps 0 --render-frames 0 --no-kernel-table"
for i in 1 2 3; do
for t in . _basetree; do
  (cd $t && python bench.py $F 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree $t', round(r['ms_per_step'],4), 'late', round(r['late_schedule']['ms_per_step'],4))")
done; done
