import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS['c2'], 'bf16', torch.device('cuda:0'), 1, 0, False)
eng2, g2, W2, ids2 = bench.make_train_engine(bench.CONFIGS['demo'], 'bf16', torch.device('cuda:0'), 1, 0, False)
l0 = None
t0 = time.time()
for i in range(3000):
    eng.train_step(from_ids=True)
    if i % 500 == 0:
        eng2.train_step(from_ids=True)
        eng.eval_step(from_ids=True)
        torch.cuda.synchronize()
        print(i, 'train loss', eng.loss(), 'demo loss', eng2.loss(), flush=True)
        l0 = l0 or eng.loss()
torch.cuda.synchronize()
print('done', time.time() - t0, 's; finite', bool(torch.isfinite(eng.params).all()), 'loss', eng.loss())
assert torch.isfinite(eng.params).all()
