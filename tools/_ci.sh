for ci in 0 247 200 171 129; do
  if [ $ci = 0 ]; then unset MG3D_SWEEP_CI; else export MG3D_SWEEP_CI=$ci; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-schedules --breakdown 2> gpurun_out/ci.err >/dev/null
  echo "CI=$ci $(grep 'kernel level 6' gpurun_out/ci.err | awk '{printf "%s %s | ", $4, $7}')"
done
