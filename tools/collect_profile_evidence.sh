#!/bin/bash
# Collects the measurements DESIGN.md / EXPERIMENTS.md quote into gpurun_out/<round>_* on the GPU box (copy what is to be
# judged into profiles/ afterwards).  One stage per call keeps a call inside gpurun's limit:
#   bash tools/collect_profile_evidence.sh r6 bench | variants | trace | loop | ngp | probes | phase | lists
R=${1:-r6}
STAGE=${2:-bench}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
QUIET="--psnr off --cpu-baseline off --render-frames 0 --ngp-steps 0 --pmc-traffic off --mapping-loop off"
line() { python3 - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
late = d.get("late_schedule") or {}
print(f'{sys.argv[1].split("/")[-1]}: {d["ms_per_step"]:.4f} ms/step, {d["value"] / 1e6:.1f} M ray-samples/s, late {late.get("ms_per_step", float("nan")):.4f}')
PY
}
case $STAGE in
bench)   # the driver's own command, every section
    python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_n1_driver_style.json 2> $OUT/${R}_bench_n1_driver_style.err
    grep "^\[bench\]" $OUT/${R}_bench_n1_driver_style.err > $OUT/${R}_bench_n1_driver_style_kernel_table.txt
    cp $OUT/live_pmc_fetch_write_per_kernel.json $OUT/${R}_pmc_fetch_write_per_kernel.json 2>/dev/null
    line $OUT/${R}_bench_n1_driver_style.json ;;
variants)
    run() { name=$1; shift; python3 $ROOT/bench.py $QUIET "$@" > $OUT/${R}_bench_$name.json 2> $OUT/${R}_bench_$name.err
            grep "^\[bench\]" $OUT/${R}_bench_$name.err > $OUT/${R}_bench_${name}_kernel_table.txt; line $OUT/${R}_bench_$name.json; }
    run n1 --steps 200 --warmup 20
    run n1_optimize_poses --steps 20 --warmup 5 --optimize-poses
    run n1_replica360 --steps 20 --warmup 5 --workload replica360
    run n1_scannet --steps 20 --warmup 5 --workload scannet
    run n1_driver_style_static128 --steps 20 --warmup 5 --static-loss-scale
    run n1_driver_style_separate_adam --steps 20 --warmup 5 --no-fuse-grid-adam
    export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
    run torchrun_world1 --steps 20 --warmup 5
    NVO_SHARD_OPT=0 run torchrun_world1_replicated_optimizer --steps 20 --warmup 5
    NVO_DIST_CAPTURE=0 run torchrun_world1_eager_collectives --steps 20 --warmup 5 ;;
trace)   # rocprofv3 kernel trace of graph-replayed steps, proposal backward beside / behind the main backward
    for ov in on off; do
        flag=""; [ $ov = off ] && flag="--no-overlap"
        rm -rf /tmp/prof_$ov
        rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$ov -- python3 $ROOT/bench.py $QUIET --steps 100 --warmup 20 --late-steps 0 --no-kernel-table $flag > /dev/null 2> $OUT/${R}_trace_$ov.err
        python3 $ROOT/tools/rocprof_clean_stats.py /tmp/prof_$ov > $OUT/${R}_bench_kernel_stats_overlap_$ov.csv 2>> $OUT/${R}_trace_$ov.err
    done
    f=$(find /tmp/prof_on -name "*kernel_trace.csv" | head -1)
    { echo "# rocprofv3 --kernel-trace of bench.py --steps 100 --warmup 20 (graph-replayed steps): every kernel of two steps with its"
      echo "# start / end relative to the step's first kernel, its duration, its HSA queue and the gap to the previous kernel of that queue"
      python3 $ROOT/tools/step_timeline.py $f; } > $OUT/${R}_step_timeline.txt
    head -12 $OUT/${R}_bench_kernel_stats_overlap_on.csv ;;
loop)    # the whole mapping run + the kernel statistics of its last 240 iterations (the trained field)
    python3 $ROOT/tools/mapping_loop.py 2> $OUT/${R}_mapping_loop.err | grep "^{" > $OUT/${R}_mapping_loop.json
    rm -rf /tmp/prof_loop
    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_loop -- python3 $ROOT/tools/mapping_loop.py --profile-steps 0 --render-frames 0 > /dev/null 2> $OUT/${R}_mapping_loop_trace.err
    python3 $ROOT/tools/rocprof_clean_stats.py /tmp/prof_loop --last 240 > $OUT/${R}_mapping_loop_last240_kernel_stats.csv 2>> $OUT/${R}_mapping_loop_trace.err
    python3 - $OUT/${R}_mapping_loop.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("loop", round(d["wall_seconds"], 3), "s,", round(d["ms_per_iteration"], 4), "ms/iteration; windows", [(w["first_iteration"], round(w["ms_per_iteration"], 4)) for w in d["windows"]],
      "render", d["render_of_the_trained_field"])
PY
    head -24 $OUT/${R}_mapping_loop_last240_kernel_stats.csv ;;
ngp)
    python3 $ROOT/tools/ngp_bench.py --steps 200 --profile --render-frames 3 --profile-render > $OUT/${R}_ngp_bench.txt 2> $OUT/${R}_ngp_bench.err
    grep -v "^{" $OUT/${R}_ngp_bench.txt | tail -45 ;;
probes)
    { for v in 1 3 4; do NVO_GRID_FWD_SMALL=$v python3 $ROOT/tools/probes/fwd_small_ab.py /tmp/ab_small.pt 2>&1 | grep -E "back to back|grid_fwd|identical"; done; } > $OUT/${R}_probe_fwd_small_forms.txt
    { echo "--- NVO_GRID_FWD_LEAN=1 (form 4 = k_grid_fwd_lean with the aligned 8-byte pair on hashed levels; form 0 = k_grid_fwd, the default)";
      NVO_GRID_FWD_LEAN=1 python3 $ROOT/tools/probes/fwd_main_ab.py 2>&1 | grep -E "form|identical"
      echo "--- NVO_GRID_FWD_LEAN=1 NVO_GRID_FWD_PAIR=0 (plain 4-byte gathers in the lean form)"
      NVO_GRID_FWD_LEAN=1 NVO_GRID_FWD_PAIR=0 python3 $ROOT/tools/probes/fwd_main_ab.py 2>&1 | grep -E "form 4"; } > $OUT/${R}_probe_fwd_main_forms.txt
    { for v in 0 1; do NVO_GRID_SLICE_CODES=$v python3 $ROOT/tools/probes/bwd_codes_ab.py /tmp/ab_codes.pt 2>&1 | grep -E "grid_bwd|identical|saved"; done; } > $OUT/${R}_probe_bwd_slice_codes.txt
    $ROOT/tools/probes/launch_probe > $OUT/${R}_probe_launch_staging.txt 2>&1
    tail -n +1 $OUT/${R}_probe_*.txt | cut -c1-170 ;;
lists)   # round 6: what a trained field's dead tiles buy (EngineConfig.sparse_backward), and what the probe costs where none are dead
    { echo "--- tools/probes/dead_tiles_ab.py (one trained state, loss scale pinned to 64): sparse steps | grid live-row list"
      NVO_AB_LOSS_SCALE=64 NVO_AB_ORDER=11,10,00 python3 $ROOT/tools/probes/dead_tiles_ab.py 2>/dev/null | grep "^rep"
      echo "--- tools/probes/bench_lists_ab.sh (driver-style workload: every tile live; auto = a sparse step every 64th step)"
      (cd $ROOT && bash tools/probes/bench_lists_ab.sh)
      echo "--- tools/probes/loop_early_ab.sh"
      (cd $ROOT && bash tools/probes/loop_early_ab.sh)
      echo "--- tools/probes/live_fraction.py"
      python3 $ROOT/tools/probes/live_fraction.py 2>/dev/null | grep "^step"; } > $OUT/${R}_probe_dead_tiles.txt
    # kernel statistics of the last 240 iterations of the whole mapping run, sparse steps armed (auto) / off (one capture set
    # per process: rocprofv3 around a process that captures a second set of step graphs crashed in hipGraphLaunch)
    for o in auto off; do
        rm -rf /tmp/pp_$o
        NVO_SPARSE_BACKWARD=$o rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_$o -- python3 $ROOT/tools/mapping_loop.py --profile-steps 0 --render-frames 0 2> /dev/null | grep "^{" > $OUT/${R}_mapping_loop_sparse_$o.json
        python3 $ROOT/tools/rocprof_clean_stats.py /tmp/pp_$o --last 240 > $OUT/${R}_trained_field_kernel_stats_sparse_$o.csv 2>/dev/null
        python3 -c "import json,sys; d=json.load(open('$OUT/${R}_mapping_loop_sparse_$o.json')); print('sparse_backward $o: loop', round(d['wall_seconds'],3), 's; windows', [(w['first_iteration'], round(w['ms_per_iteration'],4), w['loss_scale']) for w in d['windows']])" >> $OUT/${R}_probe_dead_tiles.txt
        rm -rf /tmp/pp_$o
    done
    cat $OUT/${R}_probe_dead_tiles.txt; head -14 $OUT/${R}_trained_field_kernel_stats_sparse_auto.csv ;;
phase)   # shader-clock shares of the grid kernels (an instrumented build: the product library is rebuilt afterwards)
    for v in 1 4; do
        NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE NVO_GRID_FWD_SMALL=$v python3 $ROOT/tools/grid_phase.py --steps 60 --dynamic-loss-scale 2>&1 | grep -A8 "k_grid_fwd_small, per workgroup" | sed "s/^/[NVO_GRID_FWD_SMALL=$v] /"
    done > $OUT/${R}_grid_phase_fwd_small.txt
    python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null 2>&1
    cat $OUT/${R}_grid_phase_fwd_small.txt ;;
esac
