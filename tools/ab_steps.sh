for rep in 1 2; do
for cfg in "3 1" "20 3"; do set -- $cfg
for mode in "callback 1" "poll 4"; do set -- $cfg $mode
MSK_WAIT=$3 MSK_HOST_THREADS=$4 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warm $2 wait $3 thr $4:', d['ms_per_step'], d['ms_per_step_film_in_hbm'], d['host_side']['cpu_s_per_step'])"
done; done; done
