#!/bin/bash
# Shader / memory clocks and socket power of GPU 0 while a command runs (are the matrix-core kernels' "55 % of peak" a clock question?).
# usage (through gpurun): bash tools/clock_sampler.sh <seconds> -- <command ...>
SECS=$1; shift; shift
"$@" > /dev/null 2>&1 &
PID=$!
END=$((SECONDS + SECS))
while kill -0 $PID 2>/dev/null && [ $SECONDS -lt $END ]; do
  rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' ' | sed 's/  */ /g'
  echo
  sleep 0.2
done
wait $PID
