"""One line per bench.py JSON file given: value, ms/step (helper for comparing runs).  usage: summ.py file.json ..."""
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print('==', f)
    print('value %.3e edges/s  ms/step %.2f' % (d['value'], d['ms_per_step']))
    c=d['config']; print('config', {k:c[k] for k in c if k not in ('workload','alpha','seed')})
    print('phases', {k: round(v,3) for k,v in d['phases_ms'].items()})
    print('kcore', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['kcore'].items()})
    r=d['roofline']; print('roofline', r['kernel'], 'achieved %.1f GB/s frac %.4f' % (r['achieved'], r['frac']), 'avg_launch_us %.1f' % r['avg_launch_us'], 'traffic', r.get('traffic'))
    for k,v in r['per_kernel'].items(): print('   ', k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
    if d.get('cpu_baseline'): print('cpu', d['cpu_baseline'])
