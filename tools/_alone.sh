python tools/gather_bench.py > gpurun_out/r05_g_gather_alone.txt 2>&1 || tail -5 gpurun_out/r05_g_gather_alone.txt
python tools/history_bench.py > gpurun_out/r05_g_history_alone.txt 2>&1 || tail -5 gpurun_out/r05_g_history_alone.txt
tail -5 gpurun_out/r05_g_gather_alone.txt gpurun_out/r05_g_history_alone.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
