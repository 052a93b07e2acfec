"""Cycle stamps of c1v_dec_head_fwd (library built with -DDH_STAMP for c1v_elem.hip; C1V_LIB points at it).
usage: C1V_LIB=abl/libc1dvae_dh.so python tools/dhstamp.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from conv1d_text_vae_amd import _lib
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS['c2'], 'bf16', torch.device('cuda:0'), 1, 0, False)
for _ in range(5):
    eng.train_step(from_ids=True)
torch.cuda.synchronize()
lib = _lib.load()
nwg = bench.CONFIGS['c2']['batch'] // 4
buf = (ctypes.c_ulonglong * (16 * nwg))()
lib.c1v_debug_read_dh(buf, 16 * nwg)
a = np.array(buf[:]).reshape(-1, 16).astype(np.float64)
names = ['issue loads', 'zero+barrier', 'latent', 'barrier', 'dense mfma', 'dense epi', 'barrier+copy xd', 'deconv mfma', 'park', 'barrier+up2 copy', 'drain']
print('median cycles (100 MHz s_memtime ticks x 24 = core clocks?) per phase, wave 0 of %d workgroups' % nwg)
for i, n in enumerate(names):
    print('  %-18s %8.0f' % (n, np.median(a[:, i + 1] - a[:, i])))
print('  total %.0f ; span over workgroups %.0f' % (np.median(a[:, 11] - a[:, 0]), a[:, 11].max() - a[:, 0].min()))
