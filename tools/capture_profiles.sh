#!/bin/bash
# Capture the per-round profile set on the GPU box: kernel trace + stats, PMC FETCH/WRITE passes, bench line.
# usage (through gpurun): bash tools/capture_profiles.sh r04_a [workload]      (workload: default c4 = synth-1M-20M; enron | wikipedia | reddit)
set -e
TAG=$1
WL=${2:-}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
ARGS=""
if [ -n "$WL" ]; then ARGS="--workload $WL"; TAG=${TAG}_${WL}; fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o r --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic off $ARGS > /dev/null 2>&1
echo "[capture] kernel trace done"
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
# iterations in the trace: the evaluation pre-roll (up to T = 100 batches), 12 set-up training iterations (2 launch-by-launch, 1 capture, the rest replays), 3 warm-up,
# then the 10 timed graph replays, then 10 + 5 launch-by-launch iterations that time the gather kernel (in-step, then with idle neighbours): the summary takes the
# 10 timed replays (--last 25 --count 10)
python $ROOT/tools/prof_summary.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) --top 70 --last 25 --count 10 > $OUT/${TAG}_kernel_trace_per_iter.txt
python $ROOT/tools/prof_timeline.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) --iter -20 --all > $OUT/${TAG}_timeline.txt 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o f --output-format csv -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --traffic off $ARGS > /dev/null 2>&1
echo "[capture] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p3 -o w --output-format csv -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --traffic off $ARGS > /dev/null 2>&1
echo "[capture] WRITE_SIZE pass done"
python $ROOT/tools/pmc_summary.py $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $(find /tmp/p3 -name "*counter_collection.csv" | head -1) --last 6 --out $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
# the plain line of the same box (its roofline.traffic is read from the PMC file just written when it has been copied to profiles/)
mkdir -p $ROOT/profiles && cp $OUT/${TAG}_pmc_traffic.json $ROOT/profiles/ 2>/dev/null || true
cd $ROOT && python bench.py $ARGS > $OUT/${TAG}_bench_line.json 2>/dev/null
tail -c 1500 $OUT/${TAG}_bench_line.json
