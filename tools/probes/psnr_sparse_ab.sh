# render PSNR through the reference's evaluation protocol (192 keyframes 640x480, 8192 iterations, fixed exact poses, default
# non-deterministic kernels), EngineConfig.sparse_backward auto against off, two seeds each; EXTRA=--static-loss-scale: tcnn's
# static scale of 128, under which the field hardens and the sparse steps engage
for seed in 42 43; do
for m in auto off; do
NVO_SPARSE_BACKWARD=$m python tools/eval_protocol.py --keyframes 192 --height 480 --width 640 --iterations 8192 --camera-optimizer-mode off --seed $seed --no-keyframe-views $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
e=d['evaluation_frames']
print('sparse_backward $m seed $seed: PSNR float-MSE', round(e['psnr_float_mse'],2), 'dB, reference definition', round(e.get('psnr', float('nan')),2), 'train', round(d['train_seconds'],2), 's, loss scale end', d.get('loss_scale_end'))"
done; done
