import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conv1d_text_vae_amd import engine, geometry, synthetic, _lib
V=302
g=geometry.Geometry(32,32,V,V,(128,),3,256)
W=synthetic.synthetic_vocab(8190,300)
x,ids=synthetic.synthetic_batch(W,W.shape[0],256,32,32)
eng=engine.VAEEngine(g,256,W,dtype='bf16',seed=1)
eng.load_batch(x,ids)
eng.train_step(use_graph=False)
for _ in range(3): eng.op_xent_accum()
torch.cuda.synchronize()
lib=_lib.load()
n=8*eng.x_splits*(eng.m_pad//128)
buf=(ctypes.c_ulonglong*n)()
print('rc',lib.c1v_debug_read(buf,n))
a=np.array(buf[:]).reshape(-1,8).astype(np.float64)
ctl=eng.live_ctl.cpu().numpy(); print('ctl',ctl); tiles=float(ctl[3]); a=a[:int(ctl[1])*int(ctl[2])]
names=['gemm1_a','gemm1_b','gemm2_a','gemm2_b','exp2+wait_vm','prologue(x1)','loop_total','epilogue']
for i,nm in enumerate(names):
    d = tiles if i<6 else 1
    print('%-16s mean %9.1f  min %9.1f max %9.1f (per tile)'%(nm,a[:,i].mean()/d,a[:,i].min()/d,a[:,i].max()/d) if i<6 else '%-16s mean %9.1f min %9.1f max %9.1f'%(nm,a[:,i].mean(),a[:,i].min(),a[:,i].max()))
print('tiles per wg',tiles,'(s_memtime ticks = 100 MHz? compare loop_total)')
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
for _ in range(5): eng.op_xent_accum()
e0.record()
for _ in range(20): eng.op_xent_accum()
e1.record(); torch.cuda.synchronize()
us=e0.elapsed_time(e1)/20*1e3
print('stamped kernel us',us,'=> cycles/us if loop+epilogue spans the kernel:',(a[:,6].mean()+a[:,7].mean())/us)
