mkdir -p gpurun_out/r4d
python -m pytest tests -m gpu -x -q > gpurun_out/r4d/tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r4d/tests.txt
tail -5 gpurun_out/r4d/tests.txt
python bench.py > gpurun_out/r4d/bench.txt 2>&1; tail -c 1500 gpurun_out/r4d/bench.txt
