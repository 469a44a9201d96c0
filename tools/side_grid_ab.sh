# Dev experiment (GPU box): how many workgroups should the side-set part of the step launch have?  (builds: tools/experiments/r5_side_grid_of_the_step_launch.patch)
mkdir -p gpurun_out; : > gpurun_out/side_grid_ab.txt
for rep in 1 2 3; do
for name in prod side192 side128 side96; do
  lib=$PWD/build/exp/$name/libpinn_hip.so; [ "$name" = prod ] && lib=$PWD/pinn_elastodynamics_amd/lib/libpinn_hip.so
  PINN_HIP_LIB=$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --extra-modes none --no-small-config 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$name  ms/step %.4f  launch(events) %.4f  wg0 %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['workgroup0_ms']))" >> gpurun_out/side_grid_ab.txt
done; done
sort gpurun_out/side_grid_ab.txt
