"""Which job of the step prologue is its long pole: the launch timed with single jobs removed (diagnostic)."""
import os, sys, copy, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from conv1d_text_vae_amd import _lib as L
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS[cfg], 'bf16', torch.device('cuda:0'), 1, 0, False)
for _ in range(3):
    eng.train_step(from_ids=True)
torch.cuda.synchronize()
op = eng._prologue_op(True, True, True)
base = op.c_args
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def clone(**kw):
    a = L.StepPrologueArgs.from_buffer_copy(base)
    for k, v in kw.items():
        setattr(a, k, v)
    return a
variants = {'all jobs': clone(), 'no weight pack': clone(n_pack=0), 'no rng': clone(n_rng=0), 'no gather': clone(gather=None),
            'no live rows': clone(live=None), 'no schedule': clone(nadam_state=None),
            'pack only': clone(n_rng=0, gather=None, live=None, nadam_state=None),
            'gather only': clone(n_rng=0, n_pack=0, live=None, nadam_state=None),
            'rng only': clone(n_pack=0, gather=None, live=None, nadam_state=None),
            'live only': clone(n_rng=0, n_pack=0, gather=None, nadam_state=None)}
for name, a in variants.items():
    f = lambda: eng.lib.c1v_step_prologue(C.byref(a), st)
    ms = bench._event_ms(f, 50)
    print('%-16s %.2f us' % (name, ms * 1e3))
