set -e
python -m pytest tests -m gpu -q > gpurun_out/r05_g_gpu_tests.txt 2>&1 || { tail -40 gpurun_out/r05_g_gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r05_g_gpu_tests.txt
python tools/gather_bench.py > gpurun_out/r05_g_gather_alone.txt 2>&1 || tail -5 gpurun_out/r05_g_gather_alone.txt
python tools/history_bench.py > gpurun_out/r05_g_history_alone.txt 2>&1 || tail -5 gpurun_out/r05_g_history_alone.txt
tail -4 gpurun_out/r05_g_gather_alone.txt gpurun_out/r05_g_history_alone.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
