"""Fused decoder head against the three launches at several batch sizes (C2 geometry): where the batch gate belongs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

os.environ['C1V_DEC_HEAD_MAX_BATCH'] = '100000000'
for B in (256, 512, 1024, 2048, 4096, 8192):
    cfg = dict(bench.CONFIGS['c2'], batch=B)
    row = []
    for mode in ('0', '1'):
        os.environ['C1V_DEC_HEAD_FUSE'] = mode
        eng, g, W, ids = bench.make_train_engine(cfg, 'bf16', torch.device('cuda:0'), 1, 0, False)
        eng.train_step(from_ids=True)
        ops = eng.ops_lat_train + eng.ops_dec_train[:-1]
        for _ in range(20):
            eng._list('head', ops)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 300
        e0.record()
        for _ in range(n):
            eng._list('head', ops)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / n * 1e3)
        names = [o.c_name for o in eng._lists['head'][3]]
        del eng
    print('B %5d: three launches %7.2f us, one launch %7.2f us  (%s)' % (B, row[0], row[1], names), flush=True)
