python -m pytest tests -m gpu -q > gpurun_out/r05_i_gpu_tests.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/r05_i_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
