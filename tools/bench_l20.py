import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo')
from fqtk_amd import BarcodeMatcher
rng=np.random.default_rng(5)
seen=set()
while len(seen)<384: seen.add("".join(rng.choice(list("ACGT"),size=20)))
bcs=sorted(seen)
n=100_000_000
src=rng.integers(0,384,size=n//100)
bc=np.stack([np.frombuffer(b.encode(),dtype=np.uint8) for b in bcs])[src]
obs=bc.copy(); flip=rng.random(obs.shape)<0.015; obs[flip]=np.frombuffer(b"ACGTN",dtype=np.uint8)[rng.integers(0,5,int(flip.sum()))]
d=torch.from_numpy(np.tile(obs,(100,1))).cuda()
out=torch.empty(n,dtype=torch.int32,device='cuda'); cnt=torch.zeros(385,dtype=torch.int64,device='cuda')
for kind in (2,1):
    m=BarcodeMatcher(bcs,1,2)
    if kind==1: m.memo_kind=1
    st=torch.cuda.current_stream().cuda_stream
    for _ in range(2): m.assign_batch_device(d.data_ptr(),20,n,out.data_ptr(),cnt.data_ptr(),stream=st)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): m.assign_batch_device(d.data_ptr(),20,n,out.data_ptr(),cnt.data_ptr(),stream=st)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print('L=20 S=384 kind',m.memo_kind,'%.1f G reads/s'%(n/dt/1e9), '%.0f GB/s'%(n*24/dt/1e9))
