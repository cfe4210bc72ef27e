#!/bin/bash
# Longest dependent chain of the captured training step of one workload: the runtime's DOT dump of the graph + a kernel trace of its replays
# -> tools/graph_critical_path.py.  usage (through gpurun): bash tools/critical_path.sh <tag> [workload ...]   (workload: c4 | enron | wikipedia | reddit)
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for WL in "$@"; do
ARGS=""; [ $WL != c4 ] && ARGS="--workload $WL"
rm -rf /tmp/p1 /tmp/graph_*_dot_print_*
LSTEP_GRAPH_DOT=/tmp/dot_$WL DEBUG_HIP_GRAPH_DOT_PRINT=1 python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic off $ARGS > /dev/null 2>&1 || true
cp $(ls -S /tmp/graph_*_dot_print_* | head -1) $OUT/${TAG}_graph_$WL.dot
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic off $ARGS > /dev/null 2>&1
python $ROOT/tools/prof_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) --top 90 --last 25 --count 10 > $OUT/${TAG}_${WL}_kernel_trace_per_iter.txt
python $ROOT/tools/graph_critical_path.py $OUT/${TAG}_graph_$WL.dot $OUT/${TAG}_${WL}_kernel_trace_per_iter.txt > $OUT/${TAG}_critical_path_$WL.txt 2>&1 || true
done
