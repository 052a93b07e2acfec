import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conv1d_text_vae_amd import engine, geometry, synthetic
B,T,E,N=8192,64,300,200000
V=E+2
g=geometry.Geometry(T,T,V,V,(128,),3,256)
W=synthetic.synthetic_vocab(N-2,E)
x,ids=synthetic.synthetic_batch(W,W.shape[0],B,T,T)
eng=engine.VAEEngine(g,B,W,dtype='bf16',seed=1)
eng.load_batch(x,ids)
eng.train_step(use_graph=False); torch.cuda.synchronize(); print('eager loss',eng.loss(), flush=True)
for _ in range(3): eng.train_step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): eng.train_step()
torch.cuda.synchronize()
dt=(time.perf_counter()-t0)/5
print('C3 train step ms %.2f  seq/s %.0f  loss %.4f  live frac %.3f  ctl %s'%(dt*1e3,B/dt,eng.loss(),(ids>=0).mean(),eng.live_ctl.cpu().numpy()))
fl=4.0*(ids>=0).sum()*V*N
print('loss GEMM TF/s if it were the whole step: %.0f'%(fl/dt/1e12))
