for lim in 800000 1500000 3128767 6000000 12000000 25000000; do
  KOMB_LOCAL_LIMIT=$lim timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-build 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']
print('limit $lim step %.2f peel %.2f local %.2f edges %d sweeps %d subrounds %d'%(d['ms_per_step'],p['ms_peel'],p['ms_truss_local'],d['config']['truss_local']['edges'],d['config']['truss_local']['sweeps'],d['config']['truss_subrounds']))"
done
