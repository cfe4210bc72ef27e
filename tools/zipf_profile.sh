#!/bin/bash
# Hub-skew (power-law endpoint popularity) measurement of the training step: bench line + per-iteration kernel table.
# usage (through gpurun): bash tools/zipf_profile.sh 1.2 r04_zipf12
set -e
Z=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT && python bench.py --zipf $Z --no-cpu-baseline --traffic off > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pz
rocprofv3 --kernel-trace --stats -d /tmp/pz -o r --output-format csv -- python $ROOT/bench.py --zipf $Z --steps 10 --warmup 3 --no-cpu-baseline --traffic off > /dev/null 2>&1
python $ROOT/tools/prof_summary.py $(find /tmp/pz -name "*kernel_trace.csv" | head -1) --top 40 --last 25 --count 10 > $OUT/${TAG}_kernel_trace_per_iter.txt
tail -c 400 $OUT/${TAG}_bench_line.json
