mkdir -p gpurun_out/r4e
{
timeout -k 10 600 python tests/legs_probe.py
MG3D_SWEEP_TUNE_LOG=1 python tests/legs_time.py
MG3D_LEGS=0 python tests/legs_time.py
for ci in 257 261 265; do MG3D_SWEEP_CI_32=$ci python tests/legs_time.py; done
} > gpurun_out/r4e/time.txt 2>&1
tail -60 gpurun_out/r4e/time.txt
