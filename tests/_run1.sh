mkdir -p gpurun_out/r4c
{
MG3D_SWEEP_TUNE_LOG=1 python tests/legs_time.py
for ci in 513 257 171 129 103 86; do MG3D_SWEEP_CI_32=$ci python tests/legs_time.py; done
for ci in 513 257 231 171 129; do MG3D_SWEEP_CI_40P=$ci python tests/legs_time.py; done
} > gpurun_out/r4c/time.txt 2>&1
tail -120 gpurun_out/r4c/time.txt
