for r in 1 2 3; do for spec in "$@"; do name=${spec%%=*}; lib=${spec#*=}
KOMB_ACCEL_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-build 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name step %.3f'%d['ms_per_step'])"
done; done
