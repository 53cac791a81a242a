tools/profile_round.sh r04 > gpurun_out/prof_r04.log 2>&1; tail -5 gpurun_out/prof_r04.log
tools/profile_legs.sh r04 > gpurun_out/prof_r04_legs.log 2>&1; tail -5 gpurun_out/prof_r04_legs.log
tools/profile_f32.sh r04 > gpurun_out/prof_r04_f32.log 2>&1; tail -12 gpurun_out/prof_r04_f32.log
