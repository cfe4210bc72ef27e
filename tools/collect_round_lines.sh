#!/bin/bash
# The per-round set of plain bench lines (no profiler, --traffic off: the PMC child passes belong to the main line) on one box.  usage (through gpurun): bash tools/collect_round_lines.sh r04_g
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
run() { name=$1; shift; timeout -k 10 500 python bench.py --traffic off "$@" > $OUT/${TAG}_${name}.json 2> $OUT/${TAG}_${name}.err; echo "[lines] $name: $(python -c "import json,sys; d=json.loads(open('$OUT/${TAG}_${name}.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms', round(d['value']), 'edges/s', 'frac', round(d['roofline']['frac'],3), d['roofline']['bound'], 'cpu', round(d.get('cpu_baseline',{}).get('value',0),1))" 2>&1 | tail -1)"; }
run bench_line_nograph --graph off --no-cpu-baseline
run bench_eval --mode eval --no-cpu-baseline
run bench_gap1000 --time-gap 1000 --no-cpu-baseline
for w in enron wikipedia reddit; do run ${w}_bench_line --workload $w; done
run zipf12_bench_line --zipf 1.2 --no-cpu-baseline
run zipf15_bench_line --zipf 1.5 --no-cpu-baseline
run reddit_uniform_sampler_bench_line --workload reddit --sampler uniform --history random --prime 2 --steps 3 --warmup 1 --graph off --no-cpu-baseline
run reddit_time_interval_aware_sampler_bench_line --workload reddit --sampler time_interval_aware --history random --prime 2 --steps 3 --warmup 1 --graph off --no-cpu-baseline
for f in replicate pull; do LSTEP_FORCE_DIST=1 LSTEP_FORCE_COLLECTIVES=1 LSTEP_PHASE2=$f run bench_dist_w1_rccl_$f --no-cpu-baseline; done
LSTEP_FORCE_DIST=1 LSTEP_FORCE_COLLECTIVES=1 LSTEP_PHASE2=replicate LSTEP_DIST_GRAPH=0 run bench_dist_w1_rccl_replicate_nograph --no-cpu-baseline
LSTEP_FORCE_DIST=1 LSTEP_FORCE_COLLECTIVES=1 LSTEP_PHASE2=pull LSTEP_PULL_COMM=own run bench_dist_w1_rccl_pull_own_comm --no-cpu-baseline
