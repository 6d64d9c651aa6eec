import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, kswlib, kswgen
from __graft_entry__ import load_package
pkg=load_package()
p=kswlib.make_params()
ctx=pkg.Context(0,p)
rng=np.random.default_rng(5)
pool,tasks,words=kswgen.gen_glb_realistic(rng,400)
ores,ocig=kswlib.orc_global_batch(p,pool,tasks)
for name,sel in (('w<=31',tasks['w']<=31),('32..63',(tasks['w']>31)&(tasks['w']<=63)),('>63',tasks['w']>63)):
    idx=np.nonzero(sel)[0]
    if len(idx)==0: print(name,'none'); continue
    res,cig=ctx.global_batch(pool,tasks[idx],words)
    bad=np.nonzero(res!=ores[idx])[0]
    cb=0
    for k,i in enumerate(idx):
        t=tasks[i]; o=int(t['cigar_off']); n=int(ores[i]['n_cigar'])
        if t['cigar_cap']>0 and not np.array_equal(cig[o:o+n],ocig[i]): cb+=1
    print(name,len(idx),'score/n mismatches',len(bad),'cigar mismatches',cb)
    for b in bad[:3]: print('   ',tasks[idx[b]],res[b],ores[idx[b]])
