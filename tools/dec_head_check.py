"""c1v_dec_head_fwd against the three launches it replaces: buffers after one step and parameters after 20 steps (bit for
bit), step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
res = {}
for mode in ('0', '1'):
    os.environ['C1V_DEC_HEAD_FUSE'] = mode
    eng, g, W, ids = bench.make_train_engine(bench.CONFIGS[cfg], 'bf16', torch.device('cuda:0'), 1, 0, False)
    eng.train_step(from_ids=True)
    torch.cuda.synchronize()
    bufs = {'Xd0': eng.Xd[0].t.clone(), 'Xf': eng.Xf.t.clone(), 'zd': eng.zd.clone(), 'mlv': eng.mlv.clone(),
            'z_out': eng.z_out.clone(), 'kl_b': eng.kl_b.clone(), 'Y': eng.Y.t.clone()}
    for _ in range(19):
        eng.train_step(from_ids=True)
    torch.cuda.synchronize()
    p = eng.params.clone()
    n = 2000 if cfg == 'c2' else 50
    t0 = time.perf_counter()
    for _ in range(n):
        eng.train_step(from_ids=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    rec = eng._lists.get('train@ids')
    print('fuse', mode, 'ms/step %.4f' % dt, 'launches', rec[0] if rec else None, 'loss', eng.loss(), flush=True)
    res[mode] = (bufs, p)
    del eng
for k_ in res['0'][0]:
    a, b = res['0'][0][k_], res['1'][0][k_]
    print(k_, 'identical', bool(torch.equal(a, b)), 'max diff', float((a.float() - b.float()).abs().max()))
print('params identical', bool(torch.equal(res['0'][1], res['1'][1])))
