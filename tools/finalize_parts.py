"""grad_finalize with and without the loss-scalar block (diagnostic)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from conv1d_text_vae_amd import _lib as L
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS[cfg], 'bf16', torch.device('cuda:0'), 1, 0, False)
for _ in range(2):
    eng.train_step(from_ids=True)
torch.cuda.synchronize()
op = eng.ops_finalize[-1]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
a = L.GradFinalizeArgs.from_buffer_copy(op.c_args)
b = L.GradFinalizeArgs.from_buffer_copy(op.c_args); b.loss = None
print('with loss block    %.1f us' % (bench._event_ms(lambda: eng.lib.c1v_grad_finalize_step(C.byref(a), st), 30) * 1e3))
print('without loss block %.1f us' % (bench._event_ms(lambda: eng.lib.c1v_grad_finalize_step(C.byref(b), st), 30) * 1e3))
print('blocks', a.total_blocks, 'P', eng.P, 'slab MB', sum(int(t.numel()) * 4 for t in eng.slabs.values()) / 1e6)
for n, t in eng.slabs.items():
    print('  %-50s count %9d splits %d' % (n, int(__import__("numpy").prod(eng.seg_shapes[n])), t.numel() // int(__import__("numpy").prod(eng.seg_shapes[n]))))
