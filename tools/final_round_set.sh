#!/bin/bash
# The round's final evidence on one box: profile capture (tools/capture_profiles.sh), the line as the driver runs it, the plain lines, the Zipf profile.
# usage (through gpurun): bash tools/final_round_set.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash $ROOT/tools/capture_profiles.sh r05_g > $ROOT/gpurun_out/r05_g_capture.log 2>&1
echo "[final] capture done"
cd $ROOT && python bench.py > gpurun_out/r05_g_bench_line_as_the_driver_runs_it.json 2> gpurun_out/r05_g_bench_line_as_the_driver_runs_it.err
echo "[final] driver-like line done"
bash $ROOT/tools/collect_round_lines.sh r05_g
bash $ROOT/tools/zipf_profile.sh 1.2 r05_g_zipf12 > /dev/null
echo "[final] zipf profile done"
