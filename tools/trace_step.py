import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conv1d_text_vae_amd import engine, geometry, synthetic
V=302
g=geometry.Geometry(32,32,V,V,(128,),3,256)
W=synthetic.synthetic_vocab(8190,300)
x,ids=synthetic.synthetic_batch(W,W.shape[0],256,32,32)
eng=engine.VAEEngine(g,256,W,dtype='bf16',seed=1)
eng.load_batch(x,ids)
for _ in range(6): eng.train_step()
torch.cuda.synchronize()
