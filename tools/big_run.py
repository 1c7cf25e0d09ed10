import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
host=g.load_package().host; orc=g.load_oracle()
n=int(sys.argv[1])
t0=time.time()
tasks,arena=host.synth_tasks(n,seed=77,seed_len_min=19,seed_len_max=60,seed_at_start=0,junk_frac=0.05,n_rate=0.0005)
print('gen',round(time.time()-t0,1),'s'); t0=time.time()
p=host.default_params()
with host.BswContext(device=0,streams=3,chunk_tasks=65536,pack_threads=8) as ctx:
    res=ctx.extend_pairs(p,tasks)
print('gpu submit',round(time.time()-t0,2),'s', n/(time.time()-t0)/1e6,'M seeds/s')
idx=np.arange(0,n,997)
want=orc.pair_batch(p,tasks[idx].copy(),nthreads=16)
print('parity sample', len(idx), res[idx].tobytes()==want.tobytes())
