"""Step time at C2 with the captured graph against the same launches issued one by one (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS[cfg], 'bf16', torch.device('cuda:0'), 1, 0, False)
for graph in (True, False, True):
    for _ in range(20):
        eng.train_step(use_graph=graph, from_ids=True)
    torch.cuda.synchronize()
    n = 1000
    t0 = time.perf_counter()
    for _ in range(n):
        eng.train_step(use_graph=graph, from_ids=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('graph' if graph else 'eager', 'ms/step %.4f (host issue %.4f)' % ((t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3))
