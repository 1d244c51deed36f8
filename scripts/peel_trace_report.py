import numpy as np, sys
rows=[]; sect=[]
for ln in open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/peel_trace.txt'):
    if ln.startswith('#'): sect.append(len(rows)); continue
    rows.append([float(x) for x in ln.split()])
rows=np.array(rows)
tr = rows[sect[0]:sect[1]] if len(sect)>1 else rows
print('truss steps', len(tr), 'sum ms', tr[:,8].sum()/1e3)
scan = tr[tr[:,0]==0]; proc = tr[tr[:,0]==1]
print('SCAN: n', len(scan), 'sum ms', scan[:,8].sum()/1e3, ' dense:', (scan[:,5]==0).sum(), 'sum', scan[scan[:,5]==0][:,8].sum()/1e3, ' list:', (scan[:,5]==1).sum(), 'sum', scan[scan[:,5]==1][:,8].sum()/1e3)
print('PROC: n', len(proc), 'sum ms', proc[:,8].sum()/1e3)
for lo,hi in ((0,64),(64,1024),(1024,16384),(16384,262144),(262144,4e6),(4e6,1e9)):
    w = proc[:,3]+64*proc[:,4]
    m=(w>=lo)&(w<hi)
    if m.sum(): print(f'  frontier [{lo:.0f},{hi:.0f}): n={m.sum():4d} sum={proc[m][:,8].sum()/1e3:7.2f} ms  avg={proc[m][:,8].mean():8.1f} us  avg light={proc[m][:,3].mean():10.0f} heavy={proc[m][:,4].mean():8.0f}')
print('first 14 steps:')
for r in tr[:14]: print('  mode %d L=%d light=%d heavy=%d live_mode=%d live=%d rem=%d  %.0f us' % (r[0],r[1],r[3],r[4],r[5],r[6],r[7],r[8]))
if len(sect)>1:
    co = rows[sect[1]:]
    print('core steps', len(co), 'sum ms', co[:,8].sum()/1e3, 'scan', co[co[:,0]==0][:,8].sum()/1e3, 'n', (co[:,0]==0).sum(), 'proc', co[co[:,0]==1][:,8].sum()/1e3, 'n', (co[:,0]==1).sum())
    p = co[co[:,0]==1]
    for lo,hi in ((0,64),(64,1024),(1024,16384),(16384,262144),(262144,1e9)):
        w = p[:,3]+64*p[:,4]; m=(w>=lo)&(w<hi)
        if m.sum(): print(f'  core frontier [{lo:.0f},{hi:.0f}): n={m.sum():4d} sum={p[m][:,8].sum()/1e3:7.2f} ms avg={p[m][:,8].mean():8.1f} us')
