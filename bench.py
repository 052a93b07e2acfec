#!/usr/bin/env python
"""Benchmark of the Conv1D text-VAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c4]

A "step" is one fit step (forward, cosine-softmax loss, backward, global-norm clip, Nadam, operand refresh) on one
synthetic batch that is already resident in HBM, replayed from a captured HIP graph.

  * N = 1 (default): workload = BASELINE.json configs[1]: batch 256, seq_len 32, emb 300 (+2 flag slots),
    n_filters 128, latent 256, vocabulary 8192 rows, bf16 MFMA with fp32 accumulation / fp32 master weights.  The same
    line carries sub-records: `c4` (one GPU's share of configs[3], with a per-GEMM roofline table), `f32` (C2 in fp32),
    `predict` (configs[2], C3), `predict_c5_shard` (one GPU's shard of configs[4]) and `cpu_baseline`.
  * N > 1: workload = BASELINE.json configs[3] (C4): data parallel, one process per GPU (torch.distributed, backend nccl =
    RCCL over xGMI), per-GPU batch 2048 fixed (weak scaling; global 16384 at N = 8), seq_len 128, deep 5+5 stack,
    vocabulary 32768 rows, gradients all-reduced in buckets inside the captured step.  `predict` is then configs[4] (C5):
    250000 vocabulary rows per GPU (2M at N = 8), per-shard arg-max + RCCL all-gather + merge.
    Started WITHOUT torchrun (`python bench.py --gpus N`) the parent spawns the N ranks itself (before it touches the GPU)
    and relays rank 0's line.

Rank 0 prints ONE JSON line; `roofline` is the dominant kernel measured live with HIP events, `cpu_baseline` the NumPy
oracle (a port of the reference arithmetic) timed on the host cores (N = 1 only).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1]
    'c2': dict(batch=256, seq_len=32, emb=300, n_filters=(128,), kernel_size=3, latent=256, vocab_rows=8192,
               text='C2 fit() step: Conv1dTextVAE n_filters=128 k=3 latent=256, per-GPU batch 256, seq_len 32, emb 300 '
                    '(+2 flag slots), vocabulary 8192 rows, Nadam(clipnorm=10)'),
    # BASELINE.json configs[3]: one GPU's share of the global batch 16384
    'c4': dict(batch=2048, seq_len=128, emb=300, n_filters=(128, 256, 256, 512, 512), kernel_size=3, latent=256,
               vocab_rows=32768,
               text='C4 data-parallel fit() step: Conv1dTextVAE n_filters=(128,256,256,512,512) k=3 latent=256 (5 enc + 5 dec '
                    'layers), per-GPU batch 2048 (global 16384 on 8 GPUs), seq_len 128, emb 300 (+2 flag slots), vocabulary '
                    '32768 rows, Nadam(clipnorm=10), bucketed RCCL all-reduce inside the captured step'),
}
# the reference's own demo model (demo/conv1d_text_vae_demo.py:309-312, :227, :232): n_filters=(1024, 2048), latent 700,
# batch 128; its eng-rus corpus has <= 21 tokens per sentence and ~2.3 k distinct Russian word forms
CONFIGS['demo'] = dict(batch=128, seq_len=21, emb=300, n_filters=(1024, 2048), kernel_size=3, latent=700, vocab_rows=2304,
                       text='reference demo model (demo/conv1d_text_vae_demo.py:309-312): n_filters=(1024,2048) k=3 latent=700, '
                            'batch 128, seq_len 21, emb 300 (+2 flag slots), vocabulary 2304 rows, Nadam(clipnorm=10); 24.6 M parameters')
# BASELINE.json configs[2]: transform()/predict() inference, batch 8192, seq_len 64, 200k-word vocabulary
CFG_PREDICT = dict(batch=8192, seq_len=64, emb=300, n_filters=(128,), kernel_size=3, latent=256, vocab_words=200000)
# BASELINE.json configs[4]: 2M-word vocabulary over 8 GPUs = 250000 rows per GPU, batch 512, seq_len 256
CFG_PREDICT_C5 = dict(batch=512, seq_len=256, emb=300, n_filters=(128,), kernel_size=3, latent=256, words_per_gpu=250000)
DENSE_INPUT = os.environ.get('C1V_BENCH_DENSE_INPUT', '0') == '1'     # A/B: the round-1 form of the timed input
BF16_DENSE_PEAK_TFLOPS = 2500.0       # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MATRIX_PEAK_TFLOPS = 157.3        # same guide: fp32 MFMA
HBM_PEAK_TBS = 8.0                    # same guide: HBM3E 8 TB/s spec (6.3 TB/s measured with a float4 copy)
MIN_REGION_S = 0.25                   # every timed region is stretched to at least this long (inner replays)


def gemm_flops(g, batch):
    """Algorithmic FLOPs (2*M*N*K) of every forward GEMM of the model: [(name, flops, has_dgrad)]."""
    k = g.kernel_size
    fwd = []
    c = g.input_vector_size
    for i, f in enumerate(g.n_filters):
        fwd.append(('enc_conv%d' % (i + 1), 2.0 * batch * g.enc_len_conv[i] * k * c * f, i > 0))
        c = f
    fwd.append(('heads', 2.0 * batch * g.flat_size * 2 * g.latent_dim, True))
    fwd.append(('dec_dense', 2.0 * batch * g.latent_dim * g.n_times * g.width, True))
    c = g.width
    for i, f in enumerate(g.dec_filters):
        fwd.append(('deconv%d' % (i + 1), 2.0 * batch * g.dec_in_len[i] * k * c * f, True))
        c = f
    fwd.append(('out_conv', 2.0 * batch * g.output_text_size * k * c * g.output_vector_size, True))
    return fwd


def step_flops(g, batch, n_vocab):
    """Algorithmic FLOPs of one train step (bwd = dgrad + wgrad, except no dgrad for the first conv and no wgrad for the
    constant vocabulary matrix): (model fwd + bwd, loss GEMMs fwd + bwd) over ALL rows."""
    fwd = gemm_flops(g, batch)
    model = sum(f for _, f, _ in fwd)
    bwd = sum(f * (2.0 if dg else 1.0) for _, f, dg in fwd)
    loss = 2.0 * batch * g.output_text_size * g.output_vector_size * n_vocab
    return model + bwd, 2.0 * loss


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    omp = os.environ.get('OMP_NUM_THREADS')
    if omp:
        cores = min(cores, int(omp))
    return cores


def cpu_baseline(seconds_budget=20.0):
    """Times the oracle's train step (NumPy float32, BLAS threads = host cores given to this process) on the C2 shapes."""
    import numpy as np
    from oracle import vae_oracle as O
    from conv1d_text_vae_amd import synthetic
    c = CONFIGS['c2']
    V = c['emb'] + 2
    og = O.VAEGeometry(c['seq_len'], c['seq_len'], V, V, c['n_filters'], c['kernel_size'], c['latent'])
    params = O.init_params(og, seed=0)
    W = synthetic.synthetic_vocab(c['vocab_rows'] - 2, c['emb'])
    x, ids = synthetic.synthetic_batch(W, W.shape[0], c['batch'], c['seq_len'], c['seq_len'])
    rng = np.random.RandomState(0)
    st = O.NadamState()
    n, t_total = 0, 0.0
    while True:
        eps = rng.standard_normal((c['batch'], c['latent'])).astype(np.float32)
        me = rng.uniform(size=(c['batch'], og.flat_size)) < 0.7
        md = rng.uniform(size=(c['batch'], c['latent'])) < 0.7
        t0 = time.perf_counter()
        _, grads, _ = O.train_forward_backward(og, params, W, x, ids, eps, me, md)
        O.nadam_step(params, grads, st)
        dt = time.perf_counter() - t0
        if n > 0 or dt > seconds_budget / 2:       # first step is a warm-up unless it alone eats the budget
            t_total += dt
        n += 1
        if t_total > seconds_budget or n >= 12:
            break
    timed = max(1, n - 1) if t_total > 0 and n > 1 else 1
    return dict(value=c['batch'] * timed / t_total, unit='sequences/s', cores=host_cores(), kind='port',
                sample='%d train steps of the C2 workload (batch 256, seq 32, vocab 8192) with the NumPy fp32 oracle '
                       '(CPU restatement of the reference arithmetic with dense [B,T,N] logits, not Keras)' % timed)


def _event_ms(fn, reps, warm=3):
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def pmc_traffic(cfg_name, dtype):
    """HBM bytes per launch of the dominant kernel from SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; gfx950
    correction: FETCH_SIZE counts half of a wide streaming read) whose summary is committed under profiles/.  Only the
    summary of THIS config and dtype applies; otherwise traffic is null."""
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_vocab_stream*.json')))
    for pmc in reversed(found):
        try:
            with open(pmc) as fp:
                doc = json.load(fp)
        except (OSError, ValueError):
            continue
        if doc.get('config', 'c2') != cfg_name or doc.get('dtype', 'bf16') != dtype:
            continue
        c = doc['per_launch_avg']
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            return (2.0 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0, os.path.relpath(pmc, ROOT)
    return None, None


def make_train_engine(cfg, dtype, dev, world, rank, force_dp):
    from conv1d_text_vae_amd import engine, geometry, synthetic
    V = cfg['emb'] + 2
    g = geometry.Geometry(cfg['seq_len'], cfg['seq_len'], V, V, cfg['n_filters'], cfg['kernel_size'], cfg['latent'])
    W = synthetic.synthetic_vocab(cfg['vocab_rows'] - 2, cfg['emb'])
    B = cfg['batch']
    # every rank draws its own slice of the global batch (seed keyed by the rank)
    x, ids, in_ids = synthetic.synthetic_batch(W, W.shape[0], B, cfg['seq_len'], cfg['seq_len'], seed=1234 + 7919 * rank,
                                               with_input_ids=True)
    eng = engine.VAEEngine(g, B, W, dtype=dtype, device=dev, seed=1234, world_size=world, rank=rank, force_bucketed=force_dp)
    # inputs resident in HBM, in the form fit() hands them over (estimator.py: set_input_table + in_ids, reference
    # SequenceForVAE.__getitem__ :1374-1390 on the device): the word-vector table and one row id per step; DENSE_INPUT=1
    # feeds the assembled fp32 [B, T, V] batch instead
    eng.load_batch(x if DENSE_INPUT else None, ids)
    if not DENSE_INPUT:
        import torch
        eng.set_input_table(W)
        eng.in_ids.copy_(torch.from_numpy(in_ids))
    return eng, g, W, ids


def gemm_table(eng, g, B, dtype, reps=20):
    """Every model GEMM launch of the step timed alone with HIP events (state of the last train step), forward and the
    paired dgrad+wgrad launch of backward, against the MFMA peak."""
    peak = BF16_DENSE_PEAK_TFLOPS if dtype == 'bf16' else F32_MATRIX_PEAK_TFLOPS
    fl = gemm_flops(g, B)
    fwd_ops = list(eng.ops_enc_train) + list(eng.ops_dec)
    names = [n for n, _, _ in fl]
    assert len(fwd_ops) == len(fl), (len(fwd_ops), len(fl))
    rows = []
    for (name, f, _), op in zip(fl, fwd_ops):
        ms = _event_ms(op, reps)
        rows.append({'gemm': name + ' fwd', 'gflop': f / 1e9, 'us': ms * 1e3, 'tflops': f / ms / 1e9, 'frac': f / ms / 1e9 / peak})
    # backward stages run in reverse layer order: out_conv, deconvs (last first), dec_dense, heads, enc convs (last first)
    nl, nd = len(g.n_filters), len(g.dec_filters)
    order = ['out_conv'] + ['deconv%d' % (i + 1) for i in reversed(range(nd))] + ['dec_dense', 'heads'] + \
        ['enc_conv%d' % (i + 1) for i in reversed(range(nl))]
    by_name = {n: (f, dg) for n, f, dg in fl}
    if len(eng.bwd_stages) == len(order):
        for name, (_, _, fused) in zip(order, eng.bwd_stages):
            f, dg = by_name[name]
            f = f * (2.0 if dg else 1.0)
            ms = _event_ms(lambda fused=fused: eng._run(fused), reps)
            rows.append({'gemm': name + ' bwd (dgrad+wgrad)' if dg else name + ' bwd (wgrad)', 'gflop': f / 1e9, 'us': ms * 1e3,
                         'tflops': f / ms / 1e9, 'frac': f / ms / 1e9 / peak})
    return rows


def hbm_table(eng, g, B, ids, reps=20):
    """The memory-bound kernels of the step timed alone with HIP events against the HBM peak.  `bytes` = ALGORITHMIC bytes
    of one launch (what the kernel must read and write once, from the shapes; DESIGN.md section 4 lists the formulas) --
    not counter values: those are in profiles/r03_*_ops_pmc.json (rocprofv3 FETCH_SIZE / WRITE_SIZE per launch)."""
    es = 2 if eng.code == 1 else 4
    M = B * g.output_text_size
    n_live = int((ids >= 0).sum()) if eng.compact else M
    kp, V, Vp = eng.kp, eng.V, eng.Vp
    ut_es = 2 if eng.ut_code == 1 else 4
    # vocabulary splits of a live row tile as c1v_live_rows deals them (ctl[4] whole tiles with one slab, the rest ctl[2])
    if eng.compact:
        ctl = eng.live_ctl.cpu().numpy()
        slab_rows = min(n_live, int(ctl[4]) * 128) + max(0, n_live - int(ctl[4]) * 128) * int(ctl[2])
    else:
        slab_rows = M * eng.x_splits
    dy = eng.dOf
    Xe = eng.Xe[0]
    n_in_live = int((eng.in_ids >= 0).sum().item()) if getattr(eng, 'in_ids', None) is not None else B * g.input_text_size
    rows = []

    def add(name, op, nbytes, note):
        ms = _event_ms(op, reps)
        rows.append({'kernel': name, 'bytes': float(nbytes), 'us': ms * 1e3, 'tb_per_s': nbytes / ms / 1e9,
                     'frac': nbytes / ms / 1e9 / HBM_PEAK_TBS, 'traffic': note})
    slab_bytes = sum(int(t.numel()) * 4 for t in eng.slabs.values())
    add('grad_finalize_kernel', eng.ops_finalize[-1], slab_bytes + eng.P * 4, 'wgrad slabs in, flat fp32 gradient out')
    add('nadam_kernel', [o for o in eng.update_ops() if getattr(o, '__name__', '') != 'mark_stale'][-1], eng.P * 28,
        'p, g, m, v in; p, m, v out (28 B / parameter)')
    core = eng.ops_loss_train_core
    if eng.fuse_norm:
        Xf = eng.Xf
        add('gemm_ntw_row_kernel (out_conv fwd + l2norm)', eng.ops_dec[-1],
            B * Xf.pitch * Xf.cp * es + n_live * (Vp * 4 + kp * es + 4) + M * 4 +
            ((M - n_live) * dy.cp * es if getattr(eng, 'finish_compact', False) else 0),
            'decoder activations in (every buffer row once); fp32 y, yhat, rnorm of the live rows out -- a GEMM launch, listed '
            'here because its length is its stores')
    if len(core) == 3:
        add('l2norm_rows_kernel', core[0], n_live * (Vp * 4 + kp * es + 4), 'fp32 y of the live rows in; yhat, rnorm out')
    if getattr(eng, 'finish_compact', False):
        add('xent_finish_kernel', core[-1], slab_rows * kp * ut_es + n_live * (2 * kp * es + V * 4 + 16) + n_live * dy.cp * es,
            'U slabs, yhat, target rows of W, fp32 y in; dy out -- live rows only (the zero rows of dy come from the out conv)')
    else:
        add('xent_finish_kernel', core[-1], slab_rows * kp * ut_es + n_live * (2 * kp * es + V * 4 + 8) + M * 4 + M * dy.cp * es,
            'U slabs, yhat, target rows of W, fp32 y (live rows) in; dy (all rows) out')
    if eng._input_own_launch():
        add('live_gather_kernel (live_rows + gather_input)', eng._live_gather_op(),
            M * 8 + B * g.input_text_size * 4 + n_in_live * eng.in_table.shape[1] * 4 + B * g.input_text_size * Xe.cp * es,
            'target ids in, row table out; input ids + table rows in, encoder operand out')
    else:
        add('gather_input_kernel', eng.ops_gather_in[0], B * g.input_text_size * 4 + n_in_live * eng.in_table.shape[1] * 4 +
            B * g.input_text_size * Xe.cp * es, 'input ids + table rows in, encoder operand out')
        if eng.ops_live_rows:
            add('live_rows_kernel', eng.ops_live_rows[0], M * 8, 'target ids in, row table out (latency-bound: one block)')
    return rows


def stretch(steps, per_step_s):
    """inner: how often the K timed steps are replayed so that the timed region lasts at least MIN_REGION_S."""
    return max(1, int(-(-MIN_REGION_S // max(steps * per_step_s, 1e-9))))


def train_leg(cfg_name, dtype, dev, world, rank, steps, warmup, dist=None, force_dp=False, with_gemm_table=False,
              kernel_only=False):
    """W warm-up steps, then exactly `steps` timed steps between barriers; returns the record of rank 0 (None elsewhere)."""
    import torch
    cfg = CONFIGS[cfg_name]
    eng, g, W, ids = make_train_engine(cfg, dtype, dev, world, rank, force_dp)
    B = cfg['batch']
    torch.cuda.synchronize()

    def all_reduce(t):
        # RCCL sum of one gradient bucket; asynchronous, so the next stage runs while the bucket is on the wire
        return dist.all_reduce(t, async_op=True)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if kernel_only:
        eng.train_step(use_graph=False, from_ids=not DENSE_INPUT)
        for _ in range(steps):
            eng.op_xent_accum()
        torch.cuda.synchronize()
        return None

    for _ in range(max(warmup, 2)):
        eng.train_step(all_reduce=all_reduce, from_ids=not DENSE_INPUT)
    sync()
    # The K timed steps are replayed `inner` times back to back (same barriers around the whole region) so that the region
    # lasts >= MIN_REGION_S: K = 20 steps of 0.2 ms are otherwise a 4 ms measurement.  `inner` comes from two probe steps and
    # is the same on every rank (MAX over the ranks).
    t0 = time.perf_counter()
    for _ in range(2):
        eng.train_step(all_reduce=all_reduce, from_ids=not DENSE_INPUT)
    sync()
    probe = (time.perf_counter() - t0) / 2
    if dist is not None:
        t = torch.tensor([probe], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        probe = float(t.item())
    inner = stretch(steps, probe)
    t0 = time.perf_counter()
    for _ in range(steps * inner):
        eng.train_step(all_reduce=all_reduce, from_ids=not DENSE_INPUT)
    sync()
    elapsed = (time.perf_counter() - t0) / inner
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = eng.loss()

    # dominant kernel, measured live with HIP events on the launch stream.  Rows without a target are skipped by the loss
    # stages (zero loss, zero gradient): the kernel's algorithmic FLOPs per launch are those of the LIVE rows only, so that
    # skipped work does not inflate the roofline fraction
    model_flops, loss_flops = step_flops(g, B, W.shape[0])
    live_frac = float((ids >= 0).mean()) if getattr(eng, 'compact', False) else 1.0
    loss_live = loss_flops * live_frac
    fused = eng.logits is None
    k_ms = _event_ms(eng.op_xent_accum, 50 if cfg_name == 'c2' else 5)
    peak = BF16_DENSE_PEAK_TFLOPS if dtype == 'bf16' else F32_MATRIX_PEAK_TFLOPS
    achieved = loss_live / (k_ms * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(cfg_name, dtype) if fused else (None, None)
    rec = None
    if rank == 0:
        rec = {
            'value': B * world * steps / elapsed, 'unit': 'sequences/s', 'ms_per_step': elapsed / steps * 1e3, 'steps': steps,
            'inner': inner, 'timed_region_s': elapsed * inner,
            'config': {'workload': cfg['text'], 'global_batch': B * world, 'per_gpu_batch': B, 'seq_len': cfg['seq_len'],
                       'vocab_rows': int(W.shape[0]), 'parallelism': 'dp%d' % world,
                       'input': 'dense fp32 [B, T, V] batch' if DENSE_INPUT else
                                'word-vector table + int32 row ids [B, T] (c1v_gather_input, as fit() feeds the engine)',
                       'step_gflop_dense': (model_flops + loss_flops) / 1e9,
                       'step_gflop_live': (model_flops + loss_live) / 1e9, 'final_loss': loss,
                       'grad_allreduce_bytes': int(eng.P) * 4 if world > 1 or force_dp else 0},
            'roofline': {'bound': 'mfma',
                         'kernel': ('vocab_stream_kernel<%d,ACCUM> (fused cosine-softmax loss fwd+bwd GEMMs)' % (eng.kp // 32))
                         if fused else 'c1v_cos_softmax_accum, unfused path (NT GEMM + exp2 + TN GEMM through a logits scratch)',
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'traffic': traffic, 'traffic_source': traffic_src, 'kernel_ms': k_ms,
                         'flops_per_launch': loss_live, 'live_row_fraction': live_frac},
        }
        if dtype == 'bf16' and not DENSE_INPUT:
            rec['hbm_rooflines'] = hbm_table(eng, g, B, ids)
        if with_gemm_table:
            rows = gemm_table(eng, g, B, dtype)
            rec['gemm_rooflines'] = rows
            convs = [r for r in rows if 'conv' in r['gemm']]
            big = max(convs, key=lambda r: r['gflop'])
            rec['roofline_conv'] = {'bound': 'mfma', 'kernel': 'gemm kernel of `%s`' % big['gemm'], 'achieved': big['tflops'],
                                    'peak': peak, 'unit': 'TFLOP/s', 'frac': big['frac'], 'kernel_ms': big['us'] / 1e3,
                                    'flops_per_launch': big['gflop'] * 1e9, 'traffic': None}
    del eng
    torch.cuda.empty_cache()
    return rec


def predict_leg(dev, dtype, steps=10):
    """Secondary metric of BASELINE.json ("predict sentences/sec"): encoder -> decoder -> EOS truncation -> nearest
    vocabulary word of every reconstructed step (fused MFMA GEMM + arg-max) on the C3 shapes, excluding tokenisation and
    the character-level generator stage.  Inputs resident in HBM."""
    import numpy as np
    import torch
    from conv1d_text_vae_amd import engine, geometry
    c = CFG_PREDICT
    V = c['emb'] + 2
    B, T = c['batch'], c['seq_len']
    g = geometry.Geometry(T, T, V, V, c['n_filters'], c['kernel_size'], c['latent'])
    rng = np.random.Generator(np.random.PCG64(99))
    words = rng.standard_normal((c['vocab_words'], c['emb']), dtype=np.float32)
    words /= np.linalg.norm(words, axis=1, keepdims=True)
    eng = engine.VAEEngine(g, B, None, dtype=dtype, device=dev, seed=7)
    no_early_eos(eng)
    dec = engine.VocabDecoder(words, B * T, dtype=dtype, device=dev)
    x = np.zeros((B, T, V), dtype=np.float32)
    ids = rng.integers(0, c['vocab_words'], size=(B, T - 1))
    x[:, :T - 1, :c['emb']] = words[ids]
    x[:, T - 1, V - 1] = 1.0
    eng.load_batch(x=x)
    del x

    def step():
        post = eng.reconstruct(postprocess=True)                      # [B, T, V-1] fp32, one captured HIP graph
        dec.load_rows_normalized(post, V - 1)
        return dec.run()

    step()
    torch.cuda.synchronize()
    kept = float((eng.end_idx >= T).float().mean().item())
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flops = 2.0 * B * T * c['emb'] * c['vocab_words']
    # the arg-max kernel alone on queries that look like a trained decoder's output: vocabulary rows plus noise
    planted_queries(dec, words, B * T, c['emb'], rng)
    k_ms = _event_ms(dec.run, 5, warm=1)
    return {'metric': 'predict_sentences_per_sec', 'value': B / dt, 'unit': 'sentences/s', 'ms_per_batch': dt * 1e3,
            'steps': steps, 'timed_region_s': dt * steps, 'sentences_not_truncated': kept,
            'workload': 'C3: batch 8192, seq_len 64, emb 300, 200000-word vocabulary, encoder+decoder+EOS truncation+'
                        'nearest-word GEMM/arg-max (no tokenisation, no character generator)',
            'vocab_argmax_queries': 'vocabulary rows + 10 % Gaussian noise, unit norm (no all-zero operand rows)',
            'vocab_argmax_ms': k_ms, 'vocab_argmax_tflops': flops / (k_ms * 1e-3) / 1e12,
            'vocab_argmax_frac_of_peak': flops / (k_ms * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS}


def no_early_eos(eng):
    """Random-initialised decoders declare "end of sentence" (arg-max on the last feature, reference :542-565) at a random
    early step, after which the EOS truncation zeroes the rows -- the nearest-word GEMM would then run on mostly all-zero
    queries (and zero operands hold a higher clock).  A trained model ends a sentence where the text ends: the bias of the
    EOS feature of the final convolution is lowered so that no step is truncated and every query row is a dense vector."""
    import torch
    off = eng.seg_off['deconv_decoder_embeddings/bias'] + eng.V - 1
    eng.params[off:off + 1].fill_(-1.0e4)
    eng._pack_all()
    torch.cuda.synchronize()


def planted_queries(dec, words, M, emb, rng):
    """M unit queries = vocabulary rows + 10 % noise (what a trained decoder emits), loaded into the decoder's operand."""
    import numpy as np
    import torch
    pick = rng.integers(0, words.shape[0], size=M)
    q = torch.from_numpy(words[pick]).to(dec.dev)
    q = q + 0.1 * torch.randn(q.shape, device=dec.dev, generator=torch.Generator(device=dec.dev).manual_seed(5)) / float(np.sqrt(emb))
    q = q / q.norm(dim=1, keepdim=True)
    dec.load_queries(q)


def predict_c5_leg(dev, dtype, world, rank, dist=None, steps=10):
    """BASELINE.json configs[4] (C5): batch 512, seq_len 256; the vocabulary is sharded by rows, 250000 per GPU (2M words on
    8 GPUs).  Every rank runs the (cheap) encoder/decoder on the whole batch, the fused GEMM + arg-max over ITS rows
    (global indices through n_base), then the (score, index) pairs are all-gathered over RCCL (8 bytes per query and rank)
    and merged on the device: highest score wins, ties go to the lowest global index."""
    import ctypes as C
    import numpy as np
    import torch
    from conv1d_text_vae_amd import _lib as L
    from conv1d_text_vae_amd import engine, geometry
    from conv1d_text_vae_amd.dist import shard_range
    c = CFG_PREDICT_C5
    V = c['emb'] + 2
    B, T = c['batch'], c['seq_len']
    n_total = c['words_per_gpu'] * world
    g = geometry.Geometry(T, T, V, V, c['n_filters'], c['kernel_size'], c['latent'])
    begin, end = shard_range(n_total, world, rank)
    rng = np.random.Generator(np.random.PCG64(4242 + rank))             # each rank only materialises its own rows
    words = rng.standard_normal((end - begin, c['emb']), dtype=np.float32)
    words /= np.linalg.norm(words, axis=1, keepdims=True)
    eng = engine.VAEEngine(g, B, None, dtype=dtype, device=dev, seed=7)
    no_early_eos(eng)
    dec = engine.VocabDecoder(words, B * T, dtype=dtype, device=dev, n_base=begin)
    rq = np.random.Generator(np.random.PCG64(17))                       # the same batch on every rank
    x = np.zeros((B, T, V), dtype=np.float32)
    q = rq.standard_normal((B, T - 1, c['emb']), dtype=np.float32)
    x[:, :T - 1, :c['emb']] = q / np.linalg.norm(q, axis=2, keepdims=True)
    x[:, T - 1, V - 1] = 1.0
    eng.load_batch(x=x)
    M = B * T
    lib = L.load()
    exchange = dist is not None          # (also at world size 1 under --rehearse-multi: the same calls, one shard)
    if exchange:
        all_s = torch.empty((world, M), dtype=torch.float32, device=dev)
        all_i = torch.empty((world, M), dtype=torch.int32, device=dev)
        out_s = torch.empty(M, dtype=torch.float32, device=dev)
        out_i = torch.empty(M, dtype=torch.int32, device=dev)

    def step():
        post = eng.reconstruct(postprocess=True)
        dec.load_rows_normalized(post, V - 1)
        s, i = dec.run()
        if not exchange:
            return s, i
        dist.all_gather_into_tensor(all_s, s.contiguous())
        dist.all_gather_into_tensor(all_i, i.contiguous())
        L.check(lib.c1v_argmax_merge(all_s.data_ptr(), all_i.data_ptr(), world, M, out_s.data_ptr(), out_i.data_ptr(),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'c1v_argmax_merge')
        return out_s, out_i

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = (time.perf_counter() - t0) / steps
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    planted_queries(dec, words, M, c['emb'], np.random.Generator(np.random.PCG64(23)))
    k_ms = _event_ms(dec.run, 5, warm=1)
    flops = 2.0 * M * c['emb'] * (end - begin)
    del eng, dec
    torch.cuda.empty_cache()
    return {'metric': 'predict_sentences_per_sec', 'value': B / dt, 'unit': 'sentences/s', 'ms_per_batch': dt * 1e3,
            'steps': steps, 'timed_region_s': dt * steps, 'n_gpus': world, 'vocab_words_total': n_total, 'vocab_words_per_gpu': end - begin,
            'workload': 'C5: batch 512, seq_len 256, emb 300, vocabulary sharded by rows (250000 words per GPU; 2M on 8 GPUs): '
                        'encoder+decoder+EOS truncation+per-shard GEMM/arg-max' +
                        ('+RCCL all-gather of (score, index)+merge' if exchange else ' (one shard, no exchange at 1 GPU)'),
            'vocab_argmax_ms': k_ms, 'vocab_argmax_tflops_per_gpu': flops / (k_ms * 1e-3) / 1e12,
            'vocab_argmax_frac_of_peak': flops / (k_ms * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS}


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without torchrun: start the N ranks as children (this process never touches the GPU) and
    relay rank 0's JSON line."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in proc.stdout.decode(errors='replace').splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    sys.exit(proc.returncode if proc.returncode else (0 if line else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--config', default=None, choices=sorted(CONFIGS),
                    help='workload of the timed steps (default: c2 on one GPU, c4 = the data-parallel config on several)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-predict', action='store_true', help='skip the secondary predict-throughput measurements')
    ap.add_argument('--no-extras', action='store_true', help='skip the c4 / f32 sub-records of the one-GPU line')
    ap.add_argument('--force-dp-path', action='store_true',
                    help='rehearse the multi-rank step sequence (bucketed RCCL all-reduces inside the step graph) at any world size')
    ap.add_argument('--kernel-only', action='store_true', help='only loop the dominant kernel (for rocprofv3 --pmc passes)')
    ap.add_argument('--rehearse-multi', action='store_true',
                    help='one GPU, RCCL at world size 1: run exactly the legs and the record of an N > 1 launch (C4 data-parallel '
                         'step with captured collectives, solo rate, sharded C5 predict with all-gather + merge)')
    ap.add_argument('--dry-run', action='store_true',
                    help='rehearse the launch only (spawn, rendezvous over gloo, one all-reduce, the JSON relay): no GPU work')
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn_ranks(args, sys.argv[1:])            # does not return

    import torch

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.dry_run:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group('gloo')
            t = torch.tensor([float(rank + 1)])
            dist.all_reduce(t)
            dist.barrier()
            total = float(t.item())
            dist.destroy_process_group()
        else:
            total = 1.0
        if rank == 0:
            print(json.dumps({'metric': 'dry_run', 'n_gpus': world, 'rank_sum': total}), flush=True)
        return
    if args.rehearse_multi:
        args.force_dp_path = True
    multi = world > 1 or args.rehearse_multi
    cfg_name = args.config or ('c4' if multi else 'c2')
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    dist = None
    saved_stdout = None
    if world > 1 or args.force_dp_path:
        import torch.distributed as dist
        # RCCL writes a version banner to the process's stdout when its communicator comes up; stdout must carry the one
        # JSON line only, so file descriptor 1 points at stderr until that line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if args.force_dp_path and 'RANK' not in os.environ:
            os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
        dist.init_process_group('nccl', device_id=torch.device(dev))

    if args.kernel_only:
        train_leg(cfg_name, args.dtype, dev, world, rank, args.steps, 0, dist, args.force_dp_path, kernel_only=True)
        return

    rec = train_leg(cfg_name, args.dtype, dev, world, rank, args.steps, args.warmup, dist, args.force_dp_path,
                    with_gemm_table=(cfg_name in ('c4', 'demo') and not multi))
    out = None
    if rank == 0:
        out = {'metric': 'train_sequences_per_sec', 'value': rec['value'], 'unit': 'sequences/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': rec['ms_per_step'], 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic', 'config': rec['config'],
               'roofline': rec['roofline']}
        out['inner'] = rec['inner']
        out['timed_region_s'] = rec['timed_region_s']
        for k in ('hbm_rooflines', 'gemm_rooflines', 'roofline_conv'):
            if k in rec:
                out[k] = rec[k]
    if multi:
        # the same per-GPU workload WITHOUT the collectives, all ranks at once: the one-GPU rate a scaling figure for this
        # workload has to be taken against (the N = 1 line of this benchmark is C2, not C4; its `c4` sub-record is this too)
        solo = train_leg(cfg_name, args.dtype, dev, 1, 0, max(5, args.steps // 4), 3, None, False)
        t = torch.tensor([solo['value']], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            out['config']['one_gpu_same_workload_sequences_per_sec'] = float(t.item())
        if not args.no_predict:
            p = predict_c5_leg(dev, args.dtype, world, rank, dist)
            if rank == 0:
                out['predict'] = p
    else:
        if cfg_name == 'c2' and not args.no_extras and not args.force_dp_path:
            c4 = train_leg('c4', args.dtype, dev, 1, 0, max(20, args.steps // 10), 3, None, False, with_gemm_table=True)
            out['c4'] = c4
            out['demo'] = train_leg('demo', args.dtype, dev, 1, 0, max(20, args.steps // 10), 3, None, False, with_gemm_table=True)
            other = 'f32' if args.dtype == 'bf16' else 'bf16'
            r2 = train_leg('c2', other, dev, 1, 0, max(10, args.steps // 4), 5, None, False)
            fl = r2['config']['step_gflop_live']
            out[other] = {'value': r2['value'], 'unit': 'sequences/s', 'ms_per_step': r2['ms_per_step'],
                          'step_tflops_live': fl / r2['ms_per_step'], 'roofline': r2['roofline'],
                          'workload': 'the same C2 step in %s' % other}
        if not args.no_predict:
            out['predict'] = predict_leg(dev, args.dtype)
            out['predict_c5_shard'] = predict_c5_leg(dev, args.dtype, 1, 0, None)
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
