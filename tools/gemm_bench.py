"""Time every GEMM launch of a step alone with HIP events: forward ops, and the dgrad / wgrad of every backward stage
SEPARATELY (diagnostic).  usage: python tools/gemm_bench.py c2|c4 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS[cfg], 'bf16', dev, 1, 0, False)
B = bench.CONFIGS[cfg]['batch']
for _ in range(2):
    eng.train_step(use_graph=False, from_ids=True)
torch.cuda.synchronize()
fl = bench.gemm_flops(g, B)
by_name = {n: (f, dg) for n, f, dg in fl}
fwd_ops = list(eng.ops_enc_train) + list(eng.ops_dec)
tot = 0.0
for (name, f, _), op in zip(fl, fwd_ops):
    ms = bench._event_ms(op, reps)
    tot += ms
    print('%-12s fwd   %8.1f us  %.3f' % (name, ms * 1e3, f / ms / 1e9 / 2500.0))
print('forward total %.1f us' % (tot * 1e3))
nl, nd = len(g.n_filters), len(g.dec_filters)
order = ['out_conv'] + ['deconv%d' % (i + 1) for i in reversed(range(nd))] + ['dec_dense', 'heads'] + \
    ['enc_conv%d' % (i + 1) for i in reversed(range(nl))]
tot = 0.0
for name, (main, side, fused) in zip(order, eng.bwd_stages):
    f, dg = by_name[name]
    t_d = bench._event_ms(lambda: eng._run(main), reps) if main else 0.0
    t_w = bench._event_ms(lambda: eng._run(side), reps) if side else 0.0
    t_f = bench._event_ms(lambda: eng._run(fused), reps)
    tot += t_f
    print('%-12s dgrad %8.1f us  %.3f | wgrad %8.1f us  %.3f | stage %8.1f us  %.3f' % (
        name, t_d * 1e3, (f / t_d / 1e9 / 2500.0) if t_d else 0.0, t_w * 1e3, f / t_w / 1e9 / 2500.0 if t_w else 0.0,
        t_f * 1e3, f * (2.0 if dg else 1.0) / t_f / 1e9 / 2500.0))
print('backward total %.1f us' % (tot * 1e3))
