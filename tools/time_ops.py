"""Time single ops of a built engine with HIP events (diagnostic).  usage: python tools/time_ops.py c2|c4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
dev = torch.device('cuda:0')
eng, g, W, ids = bench.make_train_engine(bench.CONFIGS[cfg], 'bf16', dev, 1, 0, False)
for _ in range(3):
    eng.train_step(from_ids=True)
torch.cuda.synchronize()
ops = {'prologue': eng._prologue_op(True, True, True), 'xent_accum': eng.op_xent_accum, 'xent_finish': eng.op_xent_finish,
       'live_rows (standalone)': eng.ops_live_rows[0] if eng.ops_live_rows else None,
       'gather (standalone)': eng.ops_gather_in[0], 'finalize': eng.ops_finalize[0] if eng.ops_finalize else None,
       'philox (standalone)': eng.ops_rng[0] if eng.ops_rng else None, 'pack weights (standalone)': eng._pack_all,
       'l2norm': eng.ops_loss_train_core[0]}
print('prologue blocks: rng %d, pack %d' % (eng._rng_blocks, eng.pack_blocks))
for name, op in ops.items():
    if op is None:
        continue
    print('%-26s %9.1f us' % (name, bench._event_ms(op, 20) * 1e3))
