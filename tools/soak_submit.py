import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
host=g.load_package().host; orc=g.load_oracle()
rng=np.random.default_rng(123)
tasks,arena=host.synth_tasks(300000,seed=3,seed_len_min=19,seed_len_max=60,seed_at_start=0,junk_frac=0.1,n_rate=0.001,indel_rate=0.01)
p=host.default_params()
want=orc.pair_batch(p,tasks,nthreads=16)
bad=0
for it in range(120):
    streams=int(rng.integers(1,9)); chunk=int(rng.choice([1,7,100,4096,20000,33333,65536,200000])); th=int(rng.integers(1,17))
    n=int(rng.choice([0,1,5,999,20000,60000,150000,300000]))
    if chunk<100 and n>2000: n=2000
    kern=int(rng.integers(0,3))
    with host.BswContext(device=0,kernel=kern,streams=streams,chunk_tasks=chunk,pack_threads=th) as ctx:
        for rep in range(int(rng.integers(1,4))):
            lo=int(rng.integers(0,300000-n+1))
            got=ctx.extend_pairs(p,tasks[lo:lo+n])
            if got.tobytes()!=want[lo:lo+n].tobytes():
                bad+=1; print('MISMATCH',it,streams,chunk,th,n,kern)
print('soak done, mismatches:',bad)
