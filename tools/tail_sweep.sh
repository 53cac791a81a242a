for cl in "9 6" "7 7" "8 7" "9 7" "6 8" "7 8" "5 7" "13 6"; do
  for t in 0 1; do
    echo -n "c,L=$cl tail=$t: "; MG3D_SWEEP_TAIL=$t ONLY=S4 CFGS=4,8,1 XCDS=3 NO_UNFUSED=1 REPS=20 python tools/sweep_bench.py $cl 2>/dev/null | awk '{print $10}'
  done
done
