#!/bin/bash
# tools/ab_bench_env.sh REPS "K=V ..." "K=V ..." ... : bench.py --steps 20 --warmup 3 (headline only) once per environment setting, interleaved REPS times;
# prints ms per step incl. copy-back | film left in HBM | CPUs busy per line.  ON THE GPU BOX.
REPS=$1; shift
for r in $(seq $REPS); do for e in "$@"; do
  env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e:', d['ms_per_step'], d['ms_per_step_film_in_hbm'], d['host_side']['cpus_busy'], d['roofline']['timed_launches'])"
done; done
