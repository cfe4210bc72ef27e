set -e
mkdir -p gpurun_out
LSTEP_RNG_TIMING=1 python -m pytest tests -m gpu -x -q -k "rng or sampler or random" > gpurun_out/rng_tests.txt 2>&1 || { tail -30 gpurun_out/rng_tests.txt; exit 1; }
tail -3 gpurun_out/rng_tests.txt
A="--workload reddit --history random --prime 2 --steps 3 --warmup 1 --graph off --no-cpu-baseline --traffic off"
LSTEP_RNG_TIMING=1 python bench.py $A --sampler uniform 2>gpurun_out/rng_uniform_new.err | grep '^{' > gpurun_out/rng_uniform_new.json
LSTEP_RNG_NUMPY_SORT=1 LSTEP_RNG_PAGEABLE=1 python bench.py $A --sampler uniform 2>/dev/null | grep '^{' > gpurun_out/rng_uniform_old.json
LSTEP_RNG_PAGEABLE=1 python bench.py $A --sampler uniform 2>/dev/null | grep '^{' > gpurun_out/rng_uniform_sortonly.json
python bench.py $A --sampler time_interval_aware 2>/dev/null | grep '^{' > gpurun_out/rng_tia_new.json
LSTEP_RNG_NUMPY_SORT=1 LSTEP_RNG_PAGEABLE=1 python bench.py $A --sampler time_interval_aware 2>/dev/null | grep '^{' > gpurun_out/rng_tia_old.json
for f in rng_uniform_new rng_uniform_old rng_uniform_sortonly rng_tia_new rng_tia_old; do python -c "
import json,sys; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],1), 'ms/step')"; done
