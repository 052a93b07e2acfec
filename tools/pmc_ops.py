"""Per-launch PMC counters / durations of the single ops of a step (diagnostic + the HBM roofline table of DESIGN.md).

  run   (under rocprofv3):  rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 tools/pmc_ops.py run c4 ORDER.json
        runs every op of the list ONCE (after two eager warm-up steps), bracketed by three `counter_add_kernel` marker
        launches on each side, and writes the op labels (one per kernel launch, in order) to ORDER.json
  parse:                    python3 tools/pmc_ops.py parse ORDER.json DIR [DIR ...]  -> JSON {label: {counter: value}}
        (counter passes: *counter_collection.csv; a --kernel-trace pass: *kernel_trace.csv gives `ns`)
"""
import csv
import ctypes as C
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def op_list(eng, g):
    nl, nd = len(g.n_filters), len(g.dec_filters)
    names = ['enc_conv%d' % (i + 1) for i in range(nl)] + ['heads', 'dec_dense'] + ['deconv%d' % (i + 1) for i in range(nd)] + ['out_conv']
    ops = []
    if eng._input_own_launch():
        ops.append(('live_rows+gather_input', eng._live_gather_op()))
    else:
        ops.append(('gather_input', eng.ops_gather_in[0]))
        if eng.ops_live_rows:
            ops.append(('live_rows', eng.ops_live_rows[0]))
    fwd = list(eng.ops_enc_train) + list(eng.ops_lat_train) + list(eng.ops_dec)
    fnames = names[:nl + 1] + ['latent_fwd'] + names[nl + 1:]
    fused = eng._fused_head(fwd)
    if len(fused) == len(fwd) - 2:                             # latent head + decoder dense + first deconv in one launch (small batches)
        i = fnames.index('latent_fwd')
        fnames = fnames[:i] + ['dec_head (latent + dec_dense + deconv1)'] + fnames[i + 3:]
        fwd = fused
    ops += [(n + ' fwd' if n != 'latent_fwd' else n, o) for n, o in zip(fnames, fwd)]
    core = eng.ops_loss_train_core
    ops += list(zip(['l2norm', 'vocab_stream', 'xent_finish'][-len(core):], core))
    order = ['out_conv'] + ['deconv%d' % (i + 1) for i in reversed(range(nd))] + ['dec_dense', 'heads'] + \
        ['enc_conv%d' % (i + 1) for i in reversed(range(nl))]
    if getattr(eng, 'bwd_batch', None) is not None:           # small batches: the dgrad chain, then all weight gradients in one launch
        for n, (main, side, fused) in zip(order, eng.bwd_stages):
            for i, o in enumerate(main):
                ops.append((n + (' dgrad' if i == 0 else ' after%d' % i), o))
        ops.append(('wgrad batch', eng.bwd_batch))
        order = []
    for n, (main, side, fused) in zip(order, eng.bwd_stages):
        if len(fused) == 1:
            ops.append((n + ' bwd pair', fused[0]))
        else:
            for o in side:
                ops.append((n + ' wgrad', o))
            for o in main:
                ops.append((n + ' dgrad', o))
    ops.append(('grad_finalize', eng.ops_finalize[-1]))
    upd = [o for o in eng.update_ops() if getattr(o, '__name__', '') != 'mark_stale']
    ops += [('nadam' if i == len(upd) - 1 else 'update%d' % i, o) for i, o in enumerate(upd)]
    return ops


def run(cfg, order_path):
    import torch
    import bench
    from conv1d_text_vae_amd import _lib as L
    eng, g, W, ids = bench.make_train_engine(bench.CONFIGS[cfg], 'bf16', torch.device('cuda:0'), 1, 0, False)
    for _ in range(2):
        eng.train_step(use_graph=False, from_ids=True)
    torch.cuda.synchronize()
    ops = op_list(eng, g)
    lib = L.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dummy = torch.zeros(1, dtype=torch.int64, device='cuda:0')

    def marker():
        for _ in range(3):
            lib.c1v_counter_add(dummy.data_ptr(), 1, st)
    marker()
    for _, o in ops:
        o()
    marker()
    torch.cuda.synchronize()
    with open(order_path, 'w') as fp:
        json.dump([n for n, _ in ops], fp)


def parse(order_path, dirs):
    labels = json.load(open(order_path))
    out = {n: {} for n in labels}
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True) + \
                glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            rows = list(csv.DictReader(open(f)))
            trace = 'Counter_Name' not in rows[0]
            disp = {}
            for r in rows:
                did = int(r['Dispatch_Id'])
                e = disp.setdefault(did, {'name': r['Kernel_Name'], 'c': {}})
                if trace:
                    e['c']['ns'] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
                else:
                    e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            ids = sorted(disp)
            mk = [i for i in ids if 'counter_add_kernel' in disp[i]['name']]
            # the measured ops lie between the last two runs of three marker launches
            assert len(mk) >= 6, 'markers not found in %s' % f
            lo, hi = mk[-4], mk[-3]
            seq = [i for i in ids if lo < i < hi]
            assert len(seq) == len(labels), (len(seq), len(labels), f)
            for n, i in zip(labels, seq):
                out[n]['kernel'] = disp[i]['name'][:60]
                out[n].update(disp[i]['c'])
    return out


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2], sys.argv[3])
    else:
        res = parse(sys.argv[2], sys.argv[3:])
        for n, c in res.items():
            fetch = 2.0 * c.get('FETCH_SIZE', 0.0) * 1024.0        # gfx950: FETCH_SIZE counts half of a wide streaming read
            write = c.get('WRITE_SIZE', 0.0) * 1024.0
            c['hbm_bytes'] = fetch + write
            if 'ns' in c and c['ns'] > 0:
                c['tb_per_s'] = (fetch + write) / c['ns'] / 1e3
        print(json.dumps(res, indent=1))
