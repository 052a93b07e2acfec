"""Host cost per launch: c1v_run_list over n copies of one op (diagnostic)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS['c2'], 'bf16', torch.device('cuda:0'), 1, 0, False)
for _ in range(3):
    eng.train_step(from_ids=True)
torch.cuda.synchronize()
flat = eng._flat(eng.train_ops(True, True, True) + eng.update_ops())[0]
names = ['prologue', 'enc', 'heads', 'latent', 'dense', 'deconv', 'out', 'stream', 'finish', 'bwd_out', 'bwd_deconv', 'bwd_dense', 'bwd_heads', 'bwd_enc', 'finalize', 'nadam']
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for nm, o in zip(names, flat):
    n = 200
    fns = (C.c_void_p * n)(*[C.cast(o.c_fn, C.c_void_p)] * n)
    args = (C.c_void_p * n)(*[C.cast(C.pointer(o.c_args), C.c_void_p)] * n)
    eng.lib.c1v_run_list(n, fns, args, st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.lib.c1v_run_list(n, fns, args, st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%-12s host %.2f us/launch, with GPU %.2f us/launch' % (nm, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
