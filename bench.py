#!/usr/bin/env python
"""Benchmark of the Conv1D text-VAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one fit step (forward, cosine-softmax loss, backward, global-norm clip, Nadam, operand refresh) on one
synthetic batch that is already resident in HBM, replayed from a captured HIP graph.  Workload = BASELINE.json
configs[1]: per-GPU batch 256, seq_len 32, emb 300 (+2 flag slots), n_filters 128, latent 256,
vocabulary 8192 rows, bf16 MFMA with fp32 accumulation / fp32 master weights.  N > 1: data parallel, one process per
GPU (torch.distributed, backend nccl = RCCL), per-GPU work fixed (weak scaling), gradients summed over xGMI.

Rank 0 prints ONE JSON line; it also carries `roofline` (dominant kernel, measured live with HIP events) and
`cpu_baseline` (the NumPy oracle = a port of the reference arithmetic, timed on the host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG = dict(batch=256, seq_len=32, emb=300, n_filters=(128,), kernel_size=3, latent=256, vocab_rows=8192)
# BASELINE.json configs[2]: transform()/predict() inference, batch 8192, seq_len 64, 200k-word vocabulary
CFG_PREDICT = dict(batch=8192, seq_len=64, emb=300, n_filters=(128,), kernel_size=3, latent=256, vocab_words=200000)
BF16_DENSE_PEAK_TFLOPS = 2500.0       # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def step_flops(g, batch, n_vocab):
    """Algorithmic FLOPs of one train step (2*M*N*K per GEMM, bwd = dgrad + wgrad except no dgrad for the
    first conv and no wgrad for the constant vocabulary matrix)."""
    k = g.kernel_size
    fwd = []
    c = g.input_vector_size
    for i, f in enumerate(g.n_filters):
        fwd.append(('enc%d' % i, 2.0 * batch * g.enc_len_conv[i] * k * c * f, i > 0))
        c = f
    fwd.append(('heads', 2.0 * batch * g.flat_size * 2 * g.latent_dim, True))
    fwd.append(('dense', 2.0 * batch * g.latent_dim * g.n_times * g.width, True))
    c = g.width
    for i, f in enumerate(g.dec_filters):
        fwd.append(('dec%d' % i, 2.0 * batch * g.dec_in_len[i] * k * c * f, True))
        c = f
    fwd.append(('out', 2.0 * batch * g.output_text_size * k * c * g.output_vector_size, True))
    model = sum(f for _, f, _ in fwd)
    bwd = sum(f * (2.0 if dg else 1.0) for _, f, dg in fwd)
    loss = 2.0 * batch * g.output_text_size * g.output_vector_size * n_vocab
    return model + bwd + 2.0 * loss, 2.0 * loss


def cpu_baseline(seconds_budget=20.0):
    """Times the oracle's train step (NumPy float32, BLAS threads = host cores given to this process) on the same shapes."""
    import numpy as np
    from oracle import vae_oracle as O
    from conv1d_text_vae_amd import synthetic
    V = CFG['emb'] + 2
    og = O.VAEGeometry(CFG['seq_len'], CFG['seq_len'], V, V, CFG['n_filters'], CFG['kernel_size'], CFG['latent'])
    params = O.init_params(og, seed=0)
    W = synthetic.synthetic_vocab(CFG['vocab_rows'] - 2, CFG['emb'])
    x, ids = synthetic.synthetic_batch(W, W.shape[0], CFG['batch'], CFG['seq_len'], CFG['seq_len'])
    rng = np.random.RandomState(0)
    st = O.NadamState()
    n, t_total = 0, 0.0
    while True:
        eps = rng.standard_normal((CFG['batch'], CFG['latent'])).astype(np.float32)
        me = rng.uniform(size=(CFG['batch'], og.flat_size)) < 0.7
        md = rng.uniform(size=(CFG['batch'], CFG['latent'])) < 0.7
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
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    omp = os.environ.get('OMP_NUM_THREADS')
    if omp:
        cores = min(cores, int(omp))
    return dict(value=CFG['batch'] * timed / t_total, unit='sequences/s', cores=cores, kind='port',
                sample='%d train steps of the C2 workload (batch 256, seq 32, vocab 8192) with the NumPy fp32 oracle '
                       '(CPU restatement of the reference arithmetic, not Keras)' % timed)


def predict_leg(dev, dtype, steps=3):
    """Secondary metric of BASELINE.json ("predict sentences/sec"): encoder -> decoder -> EOS truncation -> nearest
    vocabulary word of every reconstructed step (fused MFMA GEMM + arg-max) on the C3 shapes, excluding tokenisation and
    the character-level generator stage.  Inputs resident in HBM."""
    import numpy as np
    import torch
    from conv1d_text_vae_amd import engine, geometry, synthetic
    c = CFG_PREDICT
    V = c['emb'] + 2
    B, T = c['batch'], c['seq_len']
    g = geometry.Geometry(T, T, V, V, c['n_filters'], c['kernel_size'], c['latent'])
    rng = np.random.Generator(np.random.PCG64(99))
    words = rng.standard_normal((c['vocab_words'], c['emb']), dtype=np.float32)
    words /= np.linalg.norm(words, axis=1, keepdims=True)
    eng = engine.VAEEngine(g, B, None, dtype=dtype, device=dev, seed=7)
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
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flops = 2.0 * B * T * c['emb'] * c['vocab_words']
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dec.run()
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1)
    return {'metric': 'predict_sentences_per_sec', 'value': B / dt, 'unit': 'sentences/s', 'ms_per_batch': dt * 1e3,
            'workload': 'C3: batch 8192, seq_len 64, emb 300, 200000-word vocabulary, encoder+decoder+EOS truncation+'
                        'nearest-word GEMM/arg-max (no tokenisation, no character generator)',
            'vocab_argmax_ms': k_ms, 'vocab_argmax_tflops': flops / (k_ms * 1e-3) / 1e12}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-predict', action='store_true', help='skip the secondary predict-throughput measurement')
    ap.add_argument('--traffic', type=float, default=None,
                    help='HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass (see profiles/)')
    ap.add_argument('--force-dp-path', action='store_true',
                    help='rehearse the multi-rank step sequence (bucketed RCCL all-reduces inside the step graph) at any world size')
    ap.add_argument('--kernel-only', action='store_true', help='only loop the dominant kernel (for rocprofv3 --pmc passes)')
    args = ap.parse_args()
    if args.traffic is None:
        # HBM bytes per launch of the dominant kernel come from SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
        # gfx950 correction: FETCH_SIZE counts half of a wide streaming read) whose summary is committed under profiles/
        import glob
        found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_vocab_stream.json')))
        pmc = found[-1] if found else ''
        if os.path.isfile(pmc):
            with open(pmc) as fp:
                c = json.load(fp)['per_launch_avg']
            args.traffic = (2.0 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0

    import numpy as np
    import torch
    from conv1d_text_vae_amd import engine, geometry, synthetic

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ...'
                             % (args.gpus, args.gpus))
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

    V = CFG['emb'] + 2
    g = geometry.Geometry(CFG['seq_len'], CFG['seq_len'], V, V, CFG['n_filters'], CFG['kernel_size'], CFG['latent'])
    W = synthetic.synthetic_vocab(CFG['vocab_rows'] - 2, CFG['emb'])
    B = CFG['batch']
    x, ids = synthetic.synthetic_batch(W, W.shape[0], B * world, CFG['seq_len'], CFG['seq_len'])
    eng = engine.VAEEngine(g, B, W, dtype=args.dtype, device=dev, seed=1234, world_size=world, rank=rank,
                           force_bucketed=args.force_dp_path)
    eng.load_batch(x[rank * B:(rank + 1) * B], ids[rank * B:(rank + 1) * B])      # inputs resident in HBM
    torch.cuda.synchronize()

    def all_reduce(t):
        # RCCL sum of one gradient bucket; asynchronous, so the next graph replays while the bucket is on the wire
        return dist.all_reduce(t, async_op=True)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.kernel_only:
        eng.train_step(use_graph=False)
        for _ in range(args.steps):
            eng.op_xent_accum()
        torch.cuda.synchronize()
        return

    for _ in range(max(args.warmup, 2)):
        eng.train_step(all_reduce=all_reduce)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step(all_reduce=all_reduce)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = eng.loss()

    # dominant kernel, measured live with HIP events on the launch stream
    total_flops, loss_flops = step_flops(g, B, W.shape[0])
    # rows without a target are skipped by the loss stages (zero loss, zero gradient): the kernel's algorithmic FLOPs per
    # launch are those of the LIVE rows only, so that skipped work does not inflate the roofline fraction
    live_frac = float((ids[rank * B:(rank + 1) * B] >= 0).mean()) if getattr(eng, 'compact', False) else 1.0
    loss_flops *= live_frac
    accum = eng.op_xent_accum
    reps = 50
    for _ in range(5):
        accum()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        accum()
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / reps
    achieved = loss_flops / (k_ms * 1e-3) / 1e12

    if rank == 0:
        out = {
            'metric': 'train_sequences_per_sec', 'value': B * world * args.steps / elapsed, 'unit': 'sequences/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'C2 fit() step: Conv1dTextVAE n_filters=128 k=3 latent=256, per-GPU batch 256, seq_len 32, '
                                   'emb 300 (+2 flag slots), vocabulary 8192 rows, Nadam(clipnorm=10)',
                       'global_batch': B * world, 'seq_len': CFG['seq_len'], 'vocab_rows': int(W.shape[0]),
                       'parallelism': 'dp%d' % world, 'step_gflop': total_flops / 1e9, 'final_loss': loss},
            'roofline': {'bound': 'mfma', 'kernel': 'vocab_stream_kernel<10,ACCUM> (fused cosine-softmax loss fwd+bwd GEMMs)',
                         'achieved': achieved, 'peak': BF16_DENSE_PEAK_TFLOPS if args.dtype == 'bf16' else 157.3,
                         'unit': 'TFLOP/s',
                         'frac': achieved / (BF16_DENSE_PEAK_TFLOPS if args.dtype == 'bf16' else 157.3),
                         'traffic': args.traffic, 'kernel_ms': k_ms, 'flops_per_launch': loss_flops,
                         'live_row_fraction': live_frac},
        }
        if world == 1 and not args.no_predict:
            del eng
            torch.cuda.empty_cache()
            out['predict'] = predict_leg(dev, args.dtype)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
