"""Cycle stamps of the wide NT kernel (library built with -DNT_STAMP for c1v_gemm.hip; C1V_LIB points at it).
usage: C1V_LIB=abl/libc1dvae_stamp.so python tools/ntwstamp.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from conv1d_text_vae_amd import _lib
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS['c4'], 'bf16', torch.device('cuda:0'), 1, 0, False)
eng.train_step(use_graph=False, from_ids=True)
torch.cuda.synchronize()
lib = _lib.load()
def probe(name, op, nwg):
    for _ in range(3): op()
    torch.cuda.synchronize()
    n = 16 * min(nwg, 8192)
    buf = (ctypes.c_ulonglong * n)()
    lib.c1v_debug_read_nt(buf, n)
    a = np.array(buf[:]).reshape(-1, 16).astype(np.float64)
    d = lambda i, j: np.median(a[:, i] - a[:, j])
    print('%-10s wgs %4d | prologue %6.0f wait-first %6.0f kloop %6.0f | barrier %6.0f park %6.0f rows %6.0f drain %6.0f | total %6.0f; span all wgs %.0f' % (
        name, nwg, d(1, 0), d(7, 1), d(2, 7), d(4, 2), d(5, 4), d(6, 5), d(3, 6), d(3, 0), a[:, 3].max() - a[:, 0].min()))
fwd = list(eng.ops_enc_train) + list(eng.ops_dec)
B = 2048
NW = int(os.environ.get("NWG_SCALE", "1"))
probe("enc1", fwd[0], (B * 130 // 256 + 1) * 1 * NW)
probe('enc2', fwd[1], (B * 66 // 256 + 1) * 2)
probe('deconv4', fwd[10], (B * 58 // 256 + 1) * 2)
probe('deconv5', fwd[11], (B * 114 // 256 + 1) * 1)
probe('out', fwd[12], (B * 132 // 256 + 1) * 3)
