for mode in off SE3; do
  python tools/eval_protocol.py --keyframes 48 --height 120 --width 160 --iterations 1500 --camera-optimizer-mode $mode 2>gpurun_out/r4_proto.err | tail -1
  python tools/eval_protocol.py --keyframes 48 --height 120 --width 160 --iterations 1500 --camera-optimizer-mode $mode --pose-noise 5e-3 5e-3 2>>gpurun_out/r4_proto.err | tail -1
done
