set -e
python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/k2.log 2>&1 || (tail -40 gpurun_out/k2.log; exit 1)
tail -3 gpurun_out/k2.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke1.log 2>&1 || (tail -60 gpurun_out/smoke1.log; exit 1)
tail -5 gpurun_out/smoke1.log
