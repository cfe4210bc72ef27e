#!/bin/bash
# How the HIP runtime replays the captured step: ms per step under the runtime's launch / graph knobs (small batches are bound by the per-node
# cost of the replay: ~6.4 us per node at B = 200 on three queues, 8.5 us on one).   usage (through gpurun): bash tools/graph_launch_modes.sh [workload ...]
# Measured (round 4, gpurun_out/graph_modes*.txt): nothing beats the defaults -- HIP_FORCE_DEV_KERNARG=1 is already the default here (0: +12 %), the packet-capture flag
# changes nothing, fewer graph queues are slower (1 queue: 0.78 ms against 0.59 at B = 200), GPU_MAX_HW_QUEUES=8 is 3.5x slower, and
# DEBUG_CLR_SKIP_RELEASE_SCOPE=1 fails / hangs (left out of the list).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
MODES=("" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "ROC_USE_FGS_KERNARG=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" \
       "HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8")
if [ -n "$LSTEP_GRAPH_MODES_EXTRA" ]; then MODES+=("$LSTEP_GRAPH_MODES_EXTRA"); fi
for wl in "$@"; do
  for mode in "${MODES[@]}"; do
    args="--no-cpu-baseline"
    if [ "$wl" != "c4" ]; then args="$args --workload $wl"; fi
    ms=$(env $mode python bench.py $args 2>/dev/null | tail -1 | python -c "import sys, json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "$wl [$mode] ms_per_step=$ms"
  done
done
