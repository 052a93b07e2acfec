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
torch.cuda.synchronize()
lib=_lib.load()
def probe(name, op, nwg):
    for _ in range(3): op()
    torch.cuda.synchronize()
    n=16*min(nwg,8192)
    buf=(ctypes.c_ulonglong*n)()
    lib.c1v_debug_read_nt(buf,n)
    a=np.array(buf[:]).reshape(-1,16).astype(np.float64)
    d=lambda i,j:(a[:,i]-a[:,j]).mean()
    print('%-10s wgs %4d prologue %6.0f kloop %6.0f | barrier %6.0f lds-park %6.0f rows %6.0f store-drain %6.0f'%(
        name,nwg,d(1,0),d(2,1),d(4,2),d(5,4),d(6,5),d(3,6)))
probe('enc conv', eng.ops_enc_train[0], 120*2)
probe('heads', eng.ops_enc_train[1], 4*8)
probe('dense', eng.ops_dec[0], 4*30)
probe('deconv', eng.ops_dec[1], 68*2)
probe('out conv', eng.ops_dec[2], 128*5)
def dist(name, op, nwg):
    for _ in range(3): op()
    torch.cuda.synchronize()
    n=16*min(nwg,8192)
    buf=(ctypes.c_ulonglong*n)()
    lib.c1v_debug_read_nt(buf,n)
    a=np.array(buf[:]).reshape(-1,16).astype(np.float64)
    for nm,(i,j) in (('prologue',(1,0)),('kloop',(2,1)),('rows',(6,5))):
        k=(a[:,i]-a[:,j])
        print(name,nm,'min %d p25 %d med %d p75 %d max %d'%(k.min(),np.percentile(k,25),np.median(k),np.percentile(k,75),k.max()))
dist('dense', eng.ops_dec[0], 120)
dist('heads', eng.ops_enc_train[1], 32)
dist('enc', eng.ops_enc_train[0], 240)
def top(name, op, nwg, gm):
    for rep in range(3):
        op(); torch.cuda.synchronize()
        n=16*min(nwg,8192)
        buf=(ctypes.c_ulonglong*n)()
        lib.c1v_debug_read_nt(buf,n)
        a=np.array(buf[:]).reshape(-1,16).astype(np.float64)
        k=(a[:,6]-a[:,5])
        o=np.argsort(-k)[:6]
        print(name,'rows top:',[(int(i)%gm,int(i)//gm,int(k[i])) for i in o], 'total top', int((a[:,3]-a[:,0]).max()))
top('dense', eng.ops_dec[0], 120, 4)
top('out', eng.ops_dec[2], 640, 128)
top('deconv', eng.ops_dec[1], 136, 68)
op=eng.ops_dec[0]
op(); torch.cuda.synchronize()
n=16*120
buf=(ctypes.c_ulonglong*n)()
lib.c1v_debug_read_nt(buf,n)
a=np.array(buf[:]).reshape(-1,16).astype(np.float64)
k=(a[:,6]-a[:,5]).astype(int).reshape(30,4)
print('dense rows by (rows) x bx (cols):'); print(k.T)
st=(a[:,0]-a[:,0].min()).astype(int).reshape(30,4)
print('Xd0 pitch', eng.Xd[0].pitch, 'cp', eng.Xd[0].cp, 'ptr %x'%eng.Xd[0].t.data_ptr())
