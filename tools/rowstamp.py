"""Cycle stamps of the whole-row output convolution (library built with -DNT_STAMP for c1v_gemm.hip; C1V_LIB points at it).
usage: C1V_LIB=abl/libc1dvae_stamp.so python tools/rowstamp.py c2|c4"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from conv1d_text_vae_amd import _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS[cfg], 'bf16', torch.device('cuda:0'), 1, 0, False)
eng.train_step(use_graph=False, from_ids=True)
torch.cuda.synchronize()
lib = _lib.load()
op = eng.ops_dec[-1]
nwg = -(-(bench.CONFIGS[cfg]["batch"] * eng.Xf.pitch) // 128)
if nwg < 256: nwg = -(-(bench.CONFIGS[cfg]["batch"] * eng.Xf.pitch) // 64)
for _ in range(3): op()
torch.cuda.synchronize()
n = 16 * min(nwg, 8192)
buf = (ctypes.c_ulonglong * n)()
lib.c1v_debug_read_nt(buf, n)
a = np.array(buf[:]).reshape(-1, 16).astype(np.float64)
d = lambda i, j: np.median(a[:, i] - a[:, j])
print('%s wgs %4d | prologue %6.0f wait-first %6.0f kloop %6.0f | barrier+sumsq %6.0f half0 %6.0f half1 %6.0f drain %6.0f | total %6.0f; span all wgs %.0f' % (
    cfg, nwg, d(1, 0), d(7, 1), d(2, 7), d(4, 2), d(5, 4), d(6, 5), d(3, 6), d(3, 0), a[:, 3].max() - a[:, 0].min()))
