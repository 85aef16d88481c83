import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from atomsmm_amd import backend as B
d=np.load('tests/golden/phenol-in-water.npz', allow_pickle=True); c={k:d[k] for k in d.keys()}
alpha=np.sqrt(-np.log(2*5e-4))/1.0
n=len(c['positions'])
ctx=B.HipContext(n, c['box'])
pos=torch.as_tensor(c['positions'],device='cuda')
print('exact -107174.89815020193; positions range', c['positions'].min(), c['positions'].max())
for g in ([20,20,20],[21,21,21],[22,22,22],[24,24,24],[20,21,21],[21,21,20],[10,10,10],[11,11,11]):
    fid=ctx.pme_create(alpha, g, c['charge'])
    f=torch.zeros((n,3),dtype=torch.float64,device='cuda'); e=torch.zeros(1,dtype=torch.float64,device='cuda')
    ctx.force_eval(fid,pos,f,energy=e); ctx.check(); print(g, e.item())
