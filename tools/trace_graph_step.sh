#!/bin/bash
# Kernel trace of the graphed training step for one workload: per-iteration kernel list in launch order with gaps.
# usage (through gpurun): bash tools/trace_graph_step.sh enron r02_enron
set -e
W=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg
rocprofv3 --kernel-trace --stats -d /tmp/pg -o r --output-format csv -- python $ROOT/bench.py --steps 40 --warmup 3 --workload $W --no-cpu-baseline > $OUT/${TAG}_bench_line.json 2> /dev/null
cp $(find /tmp/pg -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python $ROOT/tools/prof_timeline.py $(find /tmp/pg -name "*kernel_trace.csv" | head -1) --iter 130 --all > $OUT/${TAG}_timeline.txt 2>&1 || true
python $ROOT/tools/prof_summary.py $(find /tmp/pg -name "*kernel_trace.csv" | head -1) --top 70 --skip 118 --count 30 > $OUT/${TAG}_kernel_trace_per_iter.txt 2>&1 || true
tail -c 600 $OUT/${TAG}_bench_line.json
