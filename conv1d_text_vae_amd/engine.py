"""Device engine of the Conv1D text VAE: owns every HBM buffer and sequences the libc1dvae kernels.

This is the MI355X replacement for what the reference obtains from Keras' `Model.fit_generator` /
`predict_on_batch` on the graphs built by `__create_model` (conv1d_text_vae.py:1185-1343): one forward, the
cosine-softmax loss, backward, global-norm clip and Nadam update per call of `train_step`, captured once into a
HIP graph and replayed.  PyTorch is used for allocation, streams, graph capture and `torch.distributed` only.

HBM layout (T = compute dtype, bf16 or fp32):
  * activations are channels-last "sequence buffers" [batch*pitch + slack, cp]: per-sample row pitch `pitch`
    (even), valid rows [off, off+len), channel pitch cp = roundup(C, 32); padding rows/channels stay zero, which
    is what implements 'same'/'full' convolution padding and lets wgrad run over the flat row index.
  * every conv layer owns an input buffer X and an output-gradient buffer dO with the same pitch.
  * parameters: one flat fp32 buffer (Keras layouts), plus flat grad / Nadam m / v buffers; per-layer operand
    copies in T ([n_out][k*cp], K-major) are refreshed by one `pack_weights` launch after each update.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from .geometry import BN_EPS, BN_MOMENTUM, Geometry, init_keras_weights, merge_heads, split_heads

KEEP_PROB = 0.7          # Dropout(0.3) :1243, :1254
TAU = 0.1                # :1319
CLIPNORM = 10.0          # :1329
NADAM_LR = 0.002         # Keras Nadam default (the estimator's `lr` is not used for the VAE, :1329)
LOG2E = 1.4426950408889634


def _pad(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _even(n: int) -> int:
    return n + (n & 1)


class Buf:
    """torch tensor + c1v_seq descriptor."""

    def __init__(self, t: torch.Tensor, batch: int, bstride: int, off: int, length: int, cp: int, c: int):
        self.t = t
        self.batch, self.bstride, self.off, self.len, self.cp, self.c = batch, bstride, off, length, cp, c
        self.code = L.F32 if t.dtype == torch.float32 else L.BF16

    def seq(self, off: Optional[int] = None, length: Optional[int] = None) -> L.Seq:
        return L.Seq(self.t.data_ptr(), self.code, self.batch, self.bstride, self.off if off is None else off,
                     self.len if length is None else length, self.cp, self.c)

    @property
    def pitch(self) -> int:
        return self.bstride // self.cp

    def valid(self) -> torch.Tensor:
        """[batch, len, c] view of the valid region (testing / export)."""
        rows = self.t[: self.batch * self.pitch * self.cp].view(self.batch, self.pitch, self.cp)
        return rows[:, self.off:self.off + self.len, :self.c]


_warned_unfused = set()


def _warn_unfused(kp: int):
    """Once per width: the fused stream kernel has no instantiation for it, so the loss / arg-max run as separate GEMMs
    through a materialised logits scratch (slower, up to 256 MB of scratch)."""
    if kp not in _warned_unfused:
        _warned_unfused.add(kp)
        import sys
        print('c1dvae: no fused vocabulary kernel for a row width of %d elements (embedding size above 318); using the '
              'unfused GEMM path through a logits scratch' % kp, file=sys.stderr)


def _null_seq() -> L.Seq:
    return L.Seq(None, 0, 0, 0, 0, 0, 0, 0)


class VAEEngine:
    def __init__(self, geom: Geometry, batch: int, vocab_vectors: Optional[np.ndarray] = None, dtype: str = 'bf16',
                 device: str = 'cuda:0', seed: int = 0, world_size: int = 1, rank: int = 0,
                 frozen_encoder: bool = False, weights: Optional[Dict[str, np.ndarray]] = None,
                 std_scale: float = 1.0, force_bucketed: bool = False):
        if not torch.cuda.is_available():
            raise L.C1VError('VAEEngine needs a GPU: the HIP kernels are the only implementation of this path.')
        self.lib = L.load()
        self.g = geom
        self.B = int(batch)
        self.world, self.rank = int(world_size), int(rank)
        self.Bg = self.B * self.world
        self.dev = torch.device(device)
        self.tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
        if dtype not in ('bf16', 'f32'):
            raise ValueError('dtype must be bf16 or f32')
        self.code = L.BF16 if dtype == 'bf16' else L.F32
        self.frozen = bool(frozen_encoder)
        self.force_bucketed = bool(force_bucketed)      # take the multi-rank step sequence even at world_size 1 (rehearsal)
        self.seed = int(seed)
        self.std_scale = float(std_scale)
        self._keep = []        # python objects that must outlive the captured graphs
        self._graphs: Dict[str, torch.cuda.CUDAGraph] = {}
        self._lists: Dict[str, object] = {}            # recorded op lists of c1v_run_list
        k = geom.kernel_size
        B = self.B
        nl = len(geom.n_filters)

        # ---------------- parameters
        self.seg_names = [n for n, _ in geom.segments()]
        self.seg_shapes = dict(geom.segments())
        self.seg_off: Dict[str, int] = {}
        o = 0
        for n, s in geom.segments():
            self.seg_off[n] = o
            o += _pad(int(np.prod(s)), 4)             # 16-byte aligned segments
        self.P = o
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=self.dev)
        self.params = z(self.P)
        self.grads = z(self.P)
        self.nadam_m = z(self.P)
        self.nadam_v = z(self.P)
        self.nadam_state = torch.tensor([0.0, 1.0, NADAM_LR], dtype=torch.float64, device=self.dev)
        self.sched = z(8)
        self.norm_out = z(1)
        self.step_ctr = torch.zeros(1, dtype=torch.int64, device=self.dev)
        # BatchNormalization state (use_batch_norm=True): per layer the moving averages (not trainable: outside the flat
        # parameter buffer), the statistics of the current batch and the two sum vectors of the forward / backward reductions
        self.bn: Dict[str, dict] = {}
        for name, ch in geom.bn_layers():
            self.bn[name] = dict(C=ch, moving_mean=z(ch), moving_var=torch.ones(ch, dtype=torch.float32, device=self.dev),
                                 mean=z(ch), var=z(ch), sums_f=z(2 * ch), sums_b=z(2 * ch), slab=z(2 * ch),
                                 partial=z(L.BN_BLOCKS * 2 * ch))
        self._dp_all_reduce = None
        self.set_weights(weights if weights is not None else init_keras_weights(geom, seed))

        # ---------------- activation buffers
        def seqbuf(batch, pitch, off, length, c, dt=None):
            cp = _pad(c, 32)
            t = z((batch * pitch + L.SLACK_ROWS) * cp, dt=dt or self.tdt)
            return Buf(t, batch, pitch * cp, off, length, cp, c)

        self.x_in = z(B, geom.input_text_size, geom.input_vector_size)        # featuriser output (fp32)
        self.ids_in = torch.full((B, geom.output_text_size), -1, dtype=torch.int32, device=self.dev)
        self.Xe: List[Buf] = []
        self.dOe: List[Buf] = []
        self.sel: List[torch.Tensor] = []
        c_in = geom.input_vector_size
        for l in range(nl):
            Lin = geom.enc_in_len[l]
            pitch = _even(Lin + k - 1)
            self.Xe.append(seqbuf(B, pitch, 0, Lin, c_in))
            self.dOe.append(seqbuf(B, pitch, k - 1, geom.enc_len_conv[l], geom.n_filters[l]))
            self.sel.append(z(B, geom.enc_len_pool[l], geom.n_filters[l], dt=torch.uint8))
            c_in = geom.n_filters[l]
        F_last = geom.n_filters[-1]
        self.flat = geom.flat_size
        self.flat_p = _pad(self.flat, 64)
        RS = L.SLACK_ROWS                                   # zero rows behind every GEMM operand (unpredicated tile loads)
        self.H = Buf(z((B + RS) * self.flat_p, dt=self.tdt), B, self.flat_p, 0, geom.enc_len_pool[-1], F_last, F_last)
        D = geom.latent_dim
        self.D = D
        self.Dp = _pad(D, 64)
        self.D2p = _pad(2 * D, 64)
        # The heads GEMM (M = batch, K = flat) has few output tiles and a long K: it runs split-K over this many workgroups per
        # tile and the latent kernel sums the partial slabs (slab 0 = self.mlv holds the complete values afterwards)
        heads_tiles = ((B + 63) // 64) * ((2 * D + 63) // 64)
        heads_kt = (self.flat_p * (2 if self.tdt != torch.float32 else 4) + 127) // 128
        self.heads_splits = int(max(1, min(int(os.environ.get('C1V_SPLITK_MAX', '8')), 256 // max(1, heads_tiles), heads_kt // 3))) \
            if os.environ.get('C1V_SPLITK', '1') == '1' else 1
        self.mlv_parts = z(self.heads_splits, B, 2 * D)
        self.mlv = self.mlv_parts[0]
        self.eps = z(B, D)
        self.eps_zero = z(B, D)
        self.keep_e = torch.ones(B, self.flat, dtype=torch.uint8, device=self.dev)
        self.keep_d = torch.ones(B, D, dtype=torch.uint8, device=self.dev)
        self.zd = z((B + RS) * self.Dp, dt=self.tdt)
        self.z_out = z(B, D)
        self.kl_b = z(B)
        self.dzd = z(B, D)
        self.dmlv = z((B + RS) * self.D2p, dt=self.tdt)
        nd = geom.n_times * geom.width
        self.nd = nd
        self.nd_p = _pad(nd, 64)
        self.dDense = Buf(z((B + RS) * self.nd_p, dt=self.tdt), B, self.nd_p, 0, geom.n_times, geom.width, geom.width)
        self.Xd: List[Buf] = []
        self.dOd: List[Buf] = []
        c_in = geom.width
        for i, f in enumerate(geom.dec_filters):
            Lin = geom.dec_in_len[i]
            pitch = _even(Lin + 3 * (k - 1))
            self.Xd.append(seqbuf(B, pitch, k - 1, Lin, c_in))
            self.dOd.append(seqbuf(B, pitch, k - 1, geom.dec_len_deconv[i], f))
            c_in = f
        To = geom.output_text_size
        self.V = geom.output_vector_size
        pitch = _even(To + 2 * (k - 1))
        self.Xf = seqbuf(B, pitch, geom.pad_left, To, c_in)
        self.dOf = seqbuf(B, pitch, k - 1, To, self.V)
        # (rows padded to whole 16-byte words: the final convolution's epilogue then stores float4 instead of float2 -- its
        # fp32 output is the largest write of the step -- and the readers take the pitch `Vp`)
        self.Vp = _pad(self.V, 4)
        self.Y = Buf(z(B * To * self.Vp + 64), B, To * self.Vp, 0, To, self.Vp, self.V)
        # use_batch_norm: the normalisation needs the convolutions' own (post-ReLU, un-pooled / un-upsampled) outputs and the
        # gradients w.r.t. the pooled / upsampled tensors as buffers of their own
        self.Ae: List[Buf] = []
        self.dPe: List[Buf] = []
        self.Ad: List[Buf] = []
        self.dUd: List[Buf] = []
        if geom.use_batch_norm:
            for l in range(nl):
                Fl = geom.n_filters[l]
                # (fp32 whatever the compute dtype: statistics and normalisation then see un-rounded conv outputs)
                self.Ae.append(seqbuf(B, _even(geom.enc_in_len[l] + k - 1), k - 1, geom.enc_len_conv[l], Fl, dt=torch.float32))
                self.dPe.append(seqbuf(B, _even(geom.enc_len_pool[l]), 0, geom.enc_len_pool[l], Fl))
            for i, f in enumerate(geom.dec_filters):
                nxt = self.Xf if i == len(geom.dec_filters) - 1 else self.Xd[i + 1]
                self.Ad.append(seqbuf(B, _even(geom.dec_in_len[i] + 3 * (k - 1)), k - 1, geom.dec_len_deconv[i], f, dt=torch.float32))
                self.dUd.append(seqbuf(B, nxt.pitch, nxt.off, nxt.len, f))
        self.post = z(B, To, self.V - 1)
        self.end_idx = torch.zeros(B, dtype=torch.int32, device=self.dev)

        # ---------------- loss
        self.M = B * To
        self.m_pad = _pad(self.M, 128)
        self.kp = _pad(self.V + 1, 64)               # at least one spare column: `ones_col` carries the softmax denominator
        self.ones_col = self.V
        self.qscale = LOG2E / TAU
        self.yhat = z((self.m_pad + RS) * self.kp, dt=self.tdt)
        self.rnorm = z(self.m_pad)
        self.xent_rows = z(self.m_pad)
        self.xent_b = z(B)
        self.loss_out = z(3)
        self.N = 0
        self.set_vocab(vocab_vectors)

        # ---------------- operand copies of the weights + gradient slabs
        self._build_weight_ops()
        self._build_ops()

    # ------------------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------------------
    def set_weights(self, keras_weights: Dict[str, np.ndarray]):
        w = merge_heads(keras_weights) if 'z_mean/kernel' in keras_weights else keras_weights
        host = np.zeros(self.P, dtype=np.float32)
        for n in self.seg_names:
            a = np.asarray(w[n], dtype=np.float32)
            if tuple(a.shape) != tuple(self.seg_shapes[n]):
                raise ValueError('weight `{0}` has shape {1}, expected {2}'.format(n, a.shape, self.seg_shapes[n]))
            host[self.seg_off[n]:self.seg_off[n] + a.size] = a.reshape(-1)
        self.params.copy_(torch.from_numpy(host))
        for name, st in self.bn.items():
            for key, attr in (('/moving_mean', 'moving_mean'), ('/moving_variance', 'moving_var')):
                if name + key in w:
                    st[attr].copy_(torch.from_numpy(np.asarray(w[name + key], dtype=np.float32)))
        if hasattr(self, '_pack_all'):
            self._pack_all()

    def get_weights(self) -> Dict[str, np.ndarray]:
        host = self.params.detach().cpu().numpy()
        w = {n: host[self.seg_off[n]:self.seg_off[n] + int(np.prod(self.seg_shapes[n]))].reshape(self.seg_shapes[n]).copy()
             for n in self.seg_names}
        for name, st in self.bn.items():
            w[name + '/moving_mean'] = st['moving_mean'].detach().cpu().numpy().copy()
            w[name + '/moving_variance'] = st['moving_var'].detach().cpu().numpy().copy()
        return split_heads(w)

    def get_grads(self) -> Dict[str, np.ndarray]:
        host = self.grads.detach().cpu().numpy()
        w = {n: host[self.seg_off[n]:self.seg_off[n] + int(np.prod(self.seg_shapes[n]))].reshape(self.seg_shapes[n]).copy()
             for n in self.seg_names}
        return split_heads(w)

    def reset_optimizer(self, lr: float = NADAM_LR):
        self.nadam_m.zero_()
        self.nadam_v.zero_()
        self.nadam_state.copy_(torch.tensor([0.0, 1.0, lr], dtype=torch.float64))

    def set_lr(self, lr: float):
        self.nadam_state[2] = lr

    def get_lr(self) -> float:
        return float(self.nadam_state[2].item())

    def set_vocab(self, vocab_vectors: Optional[np.ndarray]):
        """W = `output_word_vectors` [N, V_out] (unit rows; reference :1326 uses its transpose as a constant)."""
        if vocab_vectors is None:
            self.N = 0
            return
        W = np.asarray(vocab_vectors, dtype=np.float32)
        if W.ndim != 2 or W.shape[1] != self.V:
            raise ValueError('vocabulary matrix must be [N, {0}]'.format(self.V))
        self.N = W.shape[0]
        self.n_pad = _pad(self.N, 64)
        host = np.zeros((self.n_pad + L.SLACK_ROWS, self.kp), dtype=np.float32)
        host[:self.N, :self.V] = W
        host[:self.N, self.ones_col] = 1.0
        self.w_vocab = torch.from_numpy(host).to(self.dev).to(self.tdt).contiguous()
        fused = self.code == L.BF16 and L.fused_width(self.kp)
        tiles = self.n_pad // 64
        if self.code == L.BF16 and not fused:
            _warn_unfused(self.kp)
        if fused:
            want = max(1, (int(os.environ.get('C1V_WG_WAVES', '1')) * 256) // max(1, self.m_pad // 128))
            self.x_splits = max(1, min(tiles, want, 16))
            self.n_chunk = 0
            self.logits = None
        else:
            self.n_chunk = int(min(self.n_pad, max(1024, (1 << 28) // self.m_pad // 64 * 64)))
            chunks = (self.N + self.n_chunk - 1) // self.n_chunk
            spc = 8 if self.n_chunk >= 2048 else 1
            self.x_splits = chunks * spc
            self.logits = torch.zeros((self.n_chunk + L.SLACK_ROWS) * self.m_pad, dtype=self.tdt, device=self.dev)
        # Live-row compaction of the loss stages (rows with ids < 0 -- the padding behind the end of a text -- have zero
        # loss and zero gradient): the fused stream kernel then runs on the compacted rows and its fixed grid is re-dealt
        # on the device into more vocabulary splits per row tile (c1v_live_rows), up to max_splits slabs of `ut`.
        self.compact = fused and os.environ.get('C1V_LIVE_ROWS', '1') == '1'
        tiles_m = max(1, self.m_pad // 128)
        if self.compact and tiles_m > 256:
            # more row tiles than CUs: whole rounds of 256 row tiles run unsplit and only the remainder is cut into
            # vocabulary splits (c1v_live_rows); the pool of workgroups must hold the row tiles plus one more round
            self.x_splits = max(self.x_splits, 2)
            self.max_splits = max(self.x_splits, min(tiles, 8))
        else:
            self.max_splits = max(self.x_splits, min(tiles, 16, (4 * 256) // tiles_m)) if self.compact else self.x_splits
        self.row_pos = torch.full((self.m_pad,), -1, dtype=torch.int32, device=self.dev)
        # the inverse table (compact row -> sample row, target id): the finish step runs over the live rows only
        self.live_rows_t = torch.zeros(self.m_pad, dtype=torch.int32, device=self.dev)
        self.live_ids_t = torch.zeros(self.m_pad, dtype=torch.int32, device=self.dev)
        self.live_ctl = torch.zeros(8, dtype=torch.int32, device=self.dev)
        self.ut = torch.zeros(self.max_splits * self.kp * self.m_pad, dtype=torch.float32, device=self.dev)
        # the training pass of the fused kernel writes its partial sums as bf16 (half the bytes out of the kernel and
        # into c1v_cos_softmax_finish, which sums them in fp32); C1V_UT_F32=1 keeps fp32 slabs (A/B measurements)
        self.ut_code = L.BF16 if fused and os.environ.get('C1V_UT_F32', '0') != '1' else L.F32
        if hasattr(self, '_ops_built'):
            self._build_ops()

    # ------------------------------------------------------------------------------------------------------
    # operand copies + wgrad slabs
    # ------------------------------------------------------------------------------------------------------
    def _pslice(self, name: str) -> int:
        return self.params.data_ptr() + 4 * self.seg_off[name]

    def _build_weight_ops(self):
        g, k = self.g, self.g.kernel_size
        descs: List[L.PackDesc] = []
        self.wop: Dict[str, torch.Tensor] = {}

        self.wld: Dict[str, int] = {}
        nblk = [0]

        def add(key, src_name, n_count, taps, c_count, cp, flip, s_tap, s_c, s_n, n_groups=1, group_pad=0):
            """One operand copy.  n_groups > 1: the n axis is n_groups runs of n_count rows, each run padded to
            group_pad rows in the copy (zero rows) -- one descriptor per run."""
            ldk = _pad(taps * cp, 64)                      # rows zero-padded to whole 128-byte K tiles
            rows = n_count if n_groups == 1 else n_groups * group_pad
            t = torch.zeros((rows + 1) * ldk, dtype=self.tdt, device=self.dev)
            self.wop[key] = t
            self.wld[key] = ldk
            esz = t.element_size()
            for gi in range(n_groups):
                descs.append(L.PackDesc(self._pslice(src_name) + 4 * gi * n_count * s_n, t.data_ptr() + esz * gi * group_pad * ldk,
                                        self.code, n_count, taps, c_count, cp, flip, ldk, s_tap, s_c, s_n, nblk[0], 0))
                nblk[0] += taps * ((c_count + L.PACK_TILE - 1) // L.PACK_TILE) * ((n_count + L.PACK_TILE - 1) // L.PACK_TILE)

        c_in = g.input_vector_size
        for l, f in enumerate(g.n_filters):
            name = 'encoder_conv1d_%d/kernel' % (l + 1)            # (k, C, F)
            add('enc%d_fwd' % l, name, f, k, c_in, _pad(c_in, 32), 0, c_in * f, f, 1)
            if l > 0:
                add('enc%d_dg' % l, name, c_in, k, f, _pad(f, 32), 1, c_in * f, 1, f)
            c_in = f
        D2 = 2 * self.D
        add('heads_fwd', 'z_heads/kernel', D2, 1, self.flat, self.flat_p, 0, 0, D2, 1)       # (flat, 2D)
        add('heads_dg', 'z_heads/kernel', self.flat, 1, D2, self.D2p, 0, 0, 1, D2)
        # (D, nd), nd = n_times rows of `width` channels.  A width that is not a multiple of 4 (18 at the benchmark shape)
        # would push the reshaping epilogue onto its element-wise path (measured: 25 k cycles instead of 4 k), so the
        # forward operand pads every row of channels to width_p (zero weight rows) and the GEMM runs on n_times * width_p
        self.width_p = _pad(g.width, 4)
        if self.width_p != g.width:
            add('dense_fwd', 'deconv_decoder_dense/kernel', g.width, 1, self.D, self.Dp, 0, 0, self.nd, 1,
                n_groups=g.n_times, group_pad=self.width_p)
        else:
            add('dense_fwd', 'deconv_decoder_dense/kernel', self.nd, 1, self.D, self.Dp, 0, 0, self.nd, 1)
        add('dense_dg', 'deconv_decoder_dense/kernel', self.D, 1, self.nd, self.nd_p, 0, 0, 1, self.nd)
        c_in = g.width
        for i, f in enumerate(g.dec_filters):
            name = 'deconv_decoder_%d_deconv1d_part2/kernel' % (i + 1)      # (k, 1, F, C)
            add('dec%d_fwd' % i, name, f, k, c_in, _pad(c_in, 32), 1, f * c_in, 1, c_in)
            add('dec%d_dg' % i, name, c_in, k, f, _pad(f, 32), 0, f * c_in, c_in, 1)
            c_in = f
        name = 'deconv_decoder_embeddings/kernel'                           # (k, C, V)
        add('out_fwd', name, self.V, k, c_in, _pad(c_in, 32), 0, c_in * self.V, self.V, 1)
        add('out_dg', name, c_in, k, self.V, _pad(self.V, 32), 1, c_in * self.V, 1, self.V)

        arr = (L.PackDesc * len(descs))(*descs)
        self.pack_descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        self.n_pack = len(descs)
        self.pack_blocks = nblk[0]

        self._wop_stale = False

        def pack_all():
            L.call('c1v_pack_weights', self.pack_descs.data_ptr(), self.n_pack, self.pack_blocks, self._stream())
            self._wop_stale = False
        self._pack_all = pack_all
        pack_all()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    # ------------------------------------------------------------------------------------------------------
    # op construction
    # ------------------------------------------------------------------------------------------------------
    def _epi(self, dst: L.Seq, row_mode=L.ROW_STORE, bias=None, relu=0, reshape_w=0, act: Optional[L.Seq] = None,
             act_rowmul=1, act_scale=1.0, sel=None, sel_bstride=0, sel_cp=0, keep=None, keep_scale=1.0,
             bias_row_w=0) -> L.Epilogue:
        return L.Epilogue(bias, relu, row_mode, reshape_w, dst, act if act is not None else _null_seq(), act_rowmul,
                          act_scale, sel, sel_bstride, sel_cp, keep, keep_scale, bias_row_w=bias_row_w)

    def _bias_ptr(self, name: str) -> int:
        return self._pslice(name)

    def _op(self, fn: str, args):
        self._keep.append(args)
        f = getattr(self.lib, fn)
        ref = C.byref(args)

        def run():
            rc = f(ref, self._stream())
            if rc != 0:
                L.check(rc, fn)
        run.c_fn, run.c_args, run.c_name = f, args, fn   # what c1v_run_list needs to issue this op natively
        return run

    def _fused_head(self, ops):
        """`ops` with every run [c1v_latent_head_fwd, c1v_dense_fwd, c1v_deconv1d_fwd] that the library's fused kernel takes
        (c1v_dec_head_fwd_check: bf16, small shapes) replaced by ONE c1v_dec_head_fwd launch -- at C2 three launches of 6.4 +
        6.1 + 5.2 us that are nearly all latency.  Bit-identical.  The recorded forms of a step (list, graph) go through
        this; op lists run one by one from Python do not.  C1V_DEC_HEAD_FUSE=0: never; C1V_DEC_HEAD_MAX_BATCH: largest batch."""
        # (small batches only: a workgroup carries four samples and fetches both weight operands for them, so beyond one
        # workgroup per CU the GEMM kernels, which share a weight tile among 64 rows, are the better form)
        if os.environ.get('C1V_DEC_HEAD_FUSE', '1') == '0' or len(ops) < 3 or \
                self.B > int(os.environ.get('C1V_DEC_HEAD_MAX_BATCH', '1024')):
            return list(ops)
        key = tuple(id(o) for o in ops)
        hit = self._head_cache.get(key)
        if hit is not None:
            return hit[1]
        names = [getattr(o, 'c_name', None) for o in ops]
        out, i = [], 0
        while i < len(ops):
            if names[i:i + 3] == ['c1v_latent_head_fwd', 'c1v_dense_fwd', 'c1v_deconv1d_fwd']:
                a = L.DecHeadArgs(C.pointer(ops[i].c_args), C.pointer(ops[i + 1].c_args), C.pointer(ops[i + 2].c_args))
                if self.lib.c1v_dec_head_fwd_check(C.byref(a)) == 0:
                    out.append(self._op('c1v_dec_head_fwd', a))
                    i += 3
                    continue
            out.append(ops[i])
            i += 1
        self._head_cache[key] = (list(ops), out)             # (the key's ops are kept alive: ids stay unique)
        return out

    # ---- BatchNormalization ops (use_batch_norm=True) ------------------------------------------------------------
    def _bn_args(self, name: str, x: Buf, mode: int, grad: Optional[L.Seq] = None, dst: Optional[L.Seq] = None, sel=None,
                 sel_bstride: int = 0, sel_cp: int = 0, keep=None, train: bool = True, sel_out=None) -> L.BnArgs:
        st = self.bn[name]
        mean, var = (st['mean'], st['var']) if train else (st['moving_mean'], st['moving_var'])
        return L.BnArgs(x.seq(), mode, grad if grad is not None else _null_seq(), sel, sel_bstride, sel_cp, keep, 1.0 / KEEP_PROB,
                        mean.data_ptr(), var.data_ptr(), BN_EPS, self._pslice(name + '/gamma'), self._pslice(name + '/beta'),
                        st['sums_b'].data_ptr(), self.Bg * x.len, dst if dst is not None else _null_seq(), sel_out)

    def _bn_sync_sums(self, t: torch.Tensor):
        """Data parallel: the sums of the normalisation run over the GLOBAL batch (one small all-reduce each)."""
        if self.world > 1:
            if self._dp_all_reduce is None:
                raise L.C1VError('use_batch_norm with world_size > 1 needs the all_reduce callable of train_step')
            w = self._dp_all_reduce(t)
            if w is not None and hasattr(w, 'wait'):
                w.wait()

    def _bn_fwd_ops(self, name: str, x: Buf, mode: int, dst: L.Seq, train: bool, sel_out=None, sel_bstride=0, sel_cp=0, keep=None):
        """Ops that turn the conv output `x` into the pooled / upsampled, normalised input of the next layer."""
        st = self.bn[name]
        a = self._bn_args(name, x, mode, dst=dst, sel_bstride=sel_bstride, sel_cp=sel_cp, keep=keep, train=train, sel_out=sel_out)
        self._keep.append(a)
        f_red, f_stats, f_app = self.lib.c1v_bn_reduce, self.lib.c1v_bn_stats, self.lib.c1v_bn_apply_fwd
        ref = C.byref(a)
        n_global = self.Bg * x.len

        def reduce_():
            L.check(f_red(ref, 0, st['partial'].data_ptr(), L.BN_BLOCKS, st['sums_f'].data_ptr(), 1.0, self._stream()), 'c1v_bn_reduce')

        def sync():
            self._bn_sync_sums(st['sums_f'])
        sync.bn_sync = st['sums_f']               # (tests emulate the all-reduce of several ranks on one GPU through this)

        def stats():
            L.check(f_stats(st['sums_f'].data_ptr(), st['C'], n_global, BN_EPS, BN_MOMENTUM, st['mean'].data_ptr(), st['var'].data_ptr(),
                            st['moving_mean'].data_ptr(), st['moving_var'].data_ptr(), self._stream()), 'c1v_bn_stats')

        def apply():
            L.check(f_app(ref, self._stream()), 'c1v_bn_apply_fwd')
        return ([reduce_, sync, stats] if train else []) + [apply]

    def _bn_bwd_ops(self, name: str, x: Buf, mode: int, grad: L.Seq, dst: L.Seq, sel=None, sel_bstride=0, sel_cp=0, keep=None):
        """Ops that turn the gradient w.r.t. the pooled / upsampled tensor into the gradient at the conv's pre-activation
        (dst) and leave the gradients of gamma / beta in the layer's `slab` (beta first, then gamma)."""
        st = self.bn[name]
        a = self._bn_args(name, x, mode, grad=grad, dst=dst, sel=sel, sel_bstride=sel_bstride, sel_cp=sel_cp, keep=keep, train=True)
        self._keep.append(a)
        f_red, f_app = self.lib.c1v_bn_reduce, self.lib.c1v_bn_apply_bwd
        ref = C.byref(a)

        def reduce_():
            L.check(f_red(ref, 1, st['partial'].data_ptr(), L.BN_BLOCKS, st['sums_b'].data_ptr(), 1.0, self._stream()), 'c1v_bn_reduce')

        def share():
            # gamma / beta gradients: this rank's LOCAL sums go into the flat gradient (a device-to-device copy, taken before
            # the collective below), whose all-reduce over the ranks makes them the sums of the global batch
            st['slab'].copy_(st['sums_b'], non_blocking=True)

        def sync():
            self._bn_sync_sums(st['sums_b'])
        sync.bn_sync = st['sums_b']

        def apply():
            L.check(f_app(ref, self._stream()), 'c1v_bn_apply_bwd')
        return [reduce_, share, sync, apply]

    def _build_ops(self):
        g, k, B = self.g, self.g.kernel_size, self.B
        bn = g.use_batch_norm
        nl = len(g.n_filters)
        nd = len(g.dec_filters)
        self._keep = []
        self._graphs = {}
        self._lists = {}
        self._head_cache = {}
        op = self._op
        # ---- encoder forward (train / inference differ only in the dropout mask of the last layer)
        self._pack_in_args = L.PackInputArgs(self.x_in.data_ptr(), g.input_text_size, g.input_vector_size, self.Xe[0].seq())
        self.ops_pack_in = [op('c1v_pack_input', self._pack_in_args)]
        self._prologue_cache = {}

        def enc_fwd(train: bool):
            ops = []
            for l in range(nl):
                last = l == nl - 1
                F = g.n_filters[l]
                dstb = self.H if last else self.Xe[l + 1]
                dst = dstb.seq(length=g.enc_len_pool[l])
                keep = self.keep_e.data_ptr() if (train and last) else None
                if bn:
                    # conv (+ bias, ReLU) stored as it is; the normalisation then pools (and drops out) into the next buffer
                    e = self._epi(self.Ae[l].seq(), L.ROW_STORE, self._bias_ptr('encoder_conv1d_%d/bias' % (l + 1)), 1)
                else:
                    e = self._epi(dst, L.ROW_POOL2, self._bias_ptr('encoder_conv1d_%d/bias' % (l + 1)), 1,
                                  sel=self.sel[l].data_ptr(), sel_bstride=g.enc_len_pool[l] * F, sel_cp=F,
                                  keep=keep, keep_scale=1.0 / KEEP_PROB)
                ops.append(op('c1v_conv1d_fwd', L.ConvArgs(self.Xe[l].seq(), 0, g.enc_len_conv[l], k,
                                                           self.wop['enc%d_fwd' % l].data_ptr(), self.wld['enc%d_fwd' % l], F, e)))
                if bn:
                    ops += self._bn_fwd_ops('encoder_batchnorm_%d' % (l + 1), self.Ae[l], L.BN_POOL2, dst, train,
                                            sel_out=self.sel[l].data_ptr(), sel_bstride=g.enc_len_pool[l] * F, sel_cp=F, keep=keep)
            mlv_seq = L.Seq(self.mlv.data_ptr(), L.F32, B, 2 * self.D, 0, 1, 2 * self.D, 2 * self.D)
            e = self._epi(mlv_seq, L.ROW_STORE, self._bias_ptr('z_heads/bias'))
            ops.append(op('c1v_dense_fwd', L.DenseArgs(self.H.t.data_ptr(), self.code, B, self.flat_p, self.flat_p,
                                                       self.wop['heads_fwd'].data_ptr(), self.wld['heads_fwd'], 2 * self.D, e,
                                                       self.heads_splits, B * 2 * self.D)))
            return ops

        def latent(train: bool, sample: bool):
            a = L.LatentArgs(self.mlv.data_ptr(), B, self.D, (self.eps if sample else self.eps_zero).data_ptr(),
                             self.keep_d.data_ptr() if train else None, KEEP_PROB, self.std_scale if sample else 0.0,
                             self.zd.data_ptr(), self.code, self.Dp, self.z_out.data_ptr(), self.kl_b.data_ptr(),
                             self.dzd.data_ptr(), self.Bg, self.dmlv.data_ptr(), self.D2p,
                             self.heads_splits, B * 2 * self.D)
            return a

        self._lat_train = latent(True, True)
        self._lat_eval = latent(False, True)
        self._lat_infer = latent(False, False)

        def dec_fwd(train: bool):
            ops = []
            padded = self.width_p != g.width
            e = self._epi(self.Xd[0].seq(), L.ROW_STORE, self._bias_ptr('deconv_decoder_dense/bias'), 1, reshape_w=self.width_p,
                          bias_row_w=g.width if padded else 0)
            ops.append(op('c1v_dense_fwd', L.DenseArgs(self.zd.data_ptr(), self.code, B, self.Dp, self.Dp,
                                                       self.wop['dense_fwd'].data_ptr(), self.wld['dense_fwd'],
                                                       g.n_times * self.width_p, e)))
            for i in range(nd):
                last = i == nd - 1
                dstb = self.Xf if last else self.Xd[i + 1]
                dst_len = g.output_text_size if last else 2 * g.dec_len_deconv[i]
                if bn:
                    e = self._epi(self.Ad[i].seq(), L.ROW_STORE, self._bias_ptr('deconv_decoder_%d_deconv1d_part2/bias' % (i + 1)), 1)
                else:
                    e = self._epi(dstb.seq(length=dst_len), L.ROW_UP2,
                                  self._bias_ptr('deconv_decoder_%d_deconv1d_part2/bias' % (i + 1)), 1)
                ops.append(op('c1v_deconv1d_fwd', L.ConvArgs(self.Xd[i].seq(), 0, g.dec_len_deconv[i], k,
                                                             self.wop['dec%d_fwd' % i].data_ptr(), self.wld['dec%d_fwd' % i],
                                                             g.dec_filters[i], e)))
                if bn:
                    ops += self._bn_fwd_ops('deconv_decoder_batchnorm_%d' % (i + 1), self.Ad[i], L.BN_UP2,
                                            dstb.seq(length=dst_len), train)
            return ops

        def out_conv(norm: bool, live_only: bool, zero_dy: bool = False):
            """The output convolution (a7).  norm: its epilogue also l2-normalises the rows into the loss kernel's operand
            (C1V_ROW_NORM: one workgroup holds a whole row) -- no c1v_l2norm_rows launch, no second pass over the fp32 rows."""
            e = self._epi(self.Y.seq(), L.ROW_NORM if norm else L.ROW_STORE, self._bias_ptr('deconv_decoder_embeddings/bias'))
            if norm:
                e.nrm_yhat, e.nrm_kp, e.nrm_rnorm, e.nrm_qscale = self.yhat.data_ptr(), self.kp, self.rnorm.data_ptr(), self.qscale
                e.nrm_pos = self.row_pos.data_ptr() if self.compact else None
                e.nrm_live_only = 1 if (live_only and self.compact) else 0
                if zero_dy and self.finish_compact:
                    e.nrm_dy = self.dOf.seq()
            return op('c1v_conv1d_fwd', L.ConvArgs(self.Xf.seq(), 0, g.output_text_size, k,
                                                   self.wop['out_fwd'].data_ptr(), self.wld['out_fwd'], self.V, e))

        self.ops_gather_in = None          # built by set_input_table (device-side batch assembly)
        if getattr(self, 'in_table', None) is not None:
            self._build_gather()
        self.ops_enc_train = enc_fwd(True)
        self.ops_enc_infer = enc_fwd(False)
        self.ops_lat_train = [op('c1v_latent_head_fwd', self._lat_train)]
        self.ops_lat_eval = [op('c1v_latent_head_fwd', self._lat_eval)]
        self.ops_lat_infer = [op('c1v_latent_head_fwd', self._lat_infer)]
        # whole-row output convolution with the normalisation in its epilogue: rows of <= 320 padded columns, input channels in
        # whole 128-byte strips (C1V_FUSE_NORM=0: STORE + c1v_l2norm_rows, the round-2 sequence; results are bit-identical)
        # Not for a long reduction over few rows (the demo model: 2048 input channels = 96 K tiles, 52 row tiles): the
        # whole-row kernel then keeps a fifth of the CUs busy where the 64 x 64 tiles of the plain kernel fill the chip
        # (measured 57 us against 40 + 5 us); C1V_FUSE_NORM=1 forces it wherever it applies.
        es = 2 if self.code == L.BF16 else 4
        env_fn = os.environ.get('C1V_FUSE_NORM', 'auto')
        k_tiles = k * ((self.Xf.cp * es) // 128)
        row_tiles = -(-(B * self.Xf.pitch) // 64)
        self.fuse_norm = (self.N > 0 and self.kp <= 320 and (self.Xf.cp * es) % 128 == 0 and 2 <= k <= 9 and env_fn != '0' and
                          (env_fn == '1' or k_tiles <= 12 or row_tiles >= 192))
        body_train = dec_fwd(True)
        body_infer = dec_fwd(False) if bn else body_train
        # the finish step over the compacted rows only: the output convolution's epilogue writes the zero gradient rows of the
        # positions without a target, the live-row table comes with its inverse (C1V_FINISH_COMPACT=0: one wave per row)
        self.finish_compact = bool(self.fuse_norm and self.compact and os.environ.get('C1V_FINISH_COMPACT', '1') != '0')
        self.ops_dec_train = body_train + [out_conv(self.fuse_norm, os.environ.get('C1V_NORM_LIVE_ONLY', '1') != '0', True)]
        self.ops_dec_infer = body_infer + [out_conv(False, False)]
        self.ops_dec_eval = body_infer + [out_conv(True, False)] if self.fuse_norm else self.ops_dec_infer
        self.ops_dec = self.ops_dec_train
        self.ops_post = []
        f_eos = self.lib.c1v_eos_truncate_normalize_ld

        def run_post():
            L.check(f_eos(self.Y.t.data_ptr(), self.Vp, B, g.output_text_size, self.V, self.post.data_ptr(),
                          self.end_idx.data_ptr(), self._stream()), 'c1v_eos_truncate_normalize_ld')
        self.ops_post = [run_post]
        self._ops_built = True
        self.ops_live_first = []
        if self.N == 0:
            return

        # ---- loss
        f_live = self.lib.c1v_live_rows
        pos_ptr = self.row_pos.data_ptr() if self.compact else None
        ctl_ptr = self.live_ctl.data_ptr() if self.compact else None

        self._live_args = L.LiveRowsArgs(self.ids_in.data_ptr(), self.M, (self.m_pad // 128) * self.x_splits, self.n_pad // 64,
                                         self.max_splits, self.row_pos.data_ptr(), self.live_ctl.data_ptr(),
                                         self.live_rows_t.data_ptr() if self.finish_compact else None,
                                         self.live_ids_t.data_ptr() if self.finish_compact else None)
        self.ops_live_rows = [op('c1v_live_rows_table', self._live_args)] if self.compact else []

        def loss_ops(train: bool):
            ops = [] if self.fuse_norm else [
                op('c1v_l2norm_rows', L.L2NormArgs(self.Y.t.data_ptr(), self.M, self.V, self.Vp, self.yhat.data_ptr(),
                                                   self.code, self.kp, self.rnorm.data_ptr(), self.qscale, pos_ptr))]
            ops.append(op('c1v_cos_softmax_accum', L.XentAccumArgs(
                self.yhat.data_ptr(), self.code, self.M, self.kp, self.w_vocab.data_ptr(), self.N, self.n_pad,
                self.ut.data_ptr(), self.m_pad, self.x_splits, 1 if train else 0, self.ones_col,
                self.logits.data_ptr() if self.logits is not None else None, self.n_chunk, ctl_ptr,
                self.ut_code if train else L.F32)))
            ops.append(op('c1v_cos_softmax_finish', L.XentFinishArgs(
                self.ut.data_ptr(), self.x_splits, self.kp, self.m_pad, self.ones_col, self.yhat.data_ptr(),
                self.w_vocab.data_ptr(), self.code, self.rnorm.data_ptr(), self.qscale, self.ids_in.data_ptr(), B,
                g.output_text_size, self.V, self.Bg, self.xent_rows.data_ptr(), self.xent_b.data_ptr(),
                self.dOf.seq() if train else _null_seq(), self.Y.t.data_ptr(), self.Vp, pos_ptr, ctl_ptr,
                self.ut_code if train else L.F32,
                self.live_rows_t.data_ptr() if self.finish_compact else None,
                self.live_ids_t.data_ptr() if self.finish_compact else None)))
            core = list(ops)
            if not self.fuse_norm:                         # (fused: the output convolution already needs the positions)
                ops = self.ops_live_rows + ops
            f_red = self.lib.c1v_loss_reduce

            def red():
                L.check(f_red(self.xent_rows.data_ptr(), self.ids_in.data_ptr(), self.kl_b.data_ptr(), B,
                              g.output_text_size, self.Bg, self.loss_out.data_ptr(), self.xent_b.data_ptr(),
                              self._stream()), 'c1v_loss_reduce')
            return ops, red, core

        self.ops_loss_train, self.op_loss_reduce, self.ops_loss_train_core = loss_ops(True)
        self.ops_live_first = self.ops_live_rows if self.fuse_norm else []
        self.op_xent_accum, self.op_xent_finish = self.ops_loss_train_core[-2], self.ops_loss_train_core[-1]   # benchmarks
        self.ops_loss_eval, red_eval, core_eval = loss_ops(False)
        self.op_xent_accum_eval = core_eval[-2]
        self.ops_loss_eval = self.ops_loss_eval + [red_eval]

        # ---- backward
        segs: List[L.SegDesc] = []
        seg_names: List[str] = []
        self.slabs: Dict[str, torch.Tensor] = {}

        fblk = [0]

        def slab(name: str, splits: int, frozen=False):
            count = int(np.prod(self.seg_shapes[name]))
            t = torch.zeros(splits * count, dtype=torch.float32, device=self.dev)
            self.slabs[name] = t
            segs.append(L.SegDesc(t.data_ptr(), count, splits, 1, 0, self.seg_off[name], count, 1 if frozen else 0, fblk[0]))
            seg_names.append(name)
            fblk[0] += (count + L.CHUNK - 1) // L.CHUNK
            return t

        # sample rows per wgrad split.  Paired launches (measured at C2 in round 2): 384 < 512 < 768.  With all weight gradients
        # in one launch (c1v_wgrad_batch, below) fewer, longer workgroups are better -- two of them share a CU's ingest, and
        # every split is a slab for grad_finalize to sum: C2 step 0.1877 ms at 384 and 512, 0.1849-0.1866 at 768.  Whether a model
        # takes the batched form is decided here, from the workgroups its weight gradients would have at that split size.
        t128 = lambda n: -(-n // 128)
        conv_layers = [(B * self.Xf.pitch, g.dec_filters[-1], self.V)] + \
            [(B * self.Xd[i].pitch, g.width if i == 0 else g.dec_filters[i - 1], g.dec_filters[i]) for i in range(nd)] + \
            [(B * self.Xe[l].pitch, g.input_vector_size if l == 0 else g.n_filters[l - 1], g.n_filters[l]) for l in range(nl)]
        def est(rps):
            return sum(t128(ci) * t128(f) * k * max(1, rows // rps) for rows, ci, f in conv_layers) + \
                t128(self.D) * t128(self.nd) + t128(self.flat) * t128(2 * self.D)
        # (at most 256 workgroups = one per CU: the batched launch then takes the kernel with the four-stage operand ring, whose
        # K tiles cost their MFMAs and not a request round trip each -- C2: 0.1834 ms at 1024 rows against 0.1854 at 768 with
        # the two-stage ring and 294 workgroups; 768 rows with the deep ring would be two rounds: 0.1899)
        rows_batched = int(os.environ.get('C1V_WGRAD_ROWS_BATCHED', '0'))
        if rows_batched <= 0:
            rows_batched = next((r_ for r_ in (768, 1024, 1536) if est(r_) <= 256), 768)
        est_wgs = est(rows_batched)
        # Measured and NOT taken (C1V_WGRAD_BATCH_FUSE=1): inside such a one-workgroup-per-CU launch the taps of a
        # convolution's weight gradient sharing a workgroup (tap_group: both operands fetched once for all taps, a third of
        # the workgroups, so more and shorter row splits for the same 256 workgroups).  Bit-identical; at C2 the batched
        # launch takes 28 us that way instead of 18 (step 0.193 ms against 0.181): the fused-tap body's K tile is 3 x the
        # MFMAs behind one barrier, and with 5-8 K tiles per workgroup its longer fill and drain are not amortised.
        tg_b = 3 if k % 3 == 0 else (2 if k % 2 == 0 else 0)
        self.batch_fuse = 0
        if est_wgs <= 256 and tg_b and os.environ.get('C1V_WGRAD_BATCH_FUSE', '0') == '1':
            def est_f(rps):
                return sum(t128(ci) * t128(f) * (k // tg_b) * max(1, rows // rps) for rows, ci, f in conv_layers) + \
                    t128(self.D) * t128(self.nd) + t128(self.flat) * t128(2 * self.D)
            rf = int(os.environ.get('C1V_WGRAD_ROWS_FUSED', '0'))
            if rf <= 0:
                rf = next((r_ for r_ in (320, 384, 512, 768, 1024) if est_f(r_) <= 256), 0)
            if rf > 0:
                self.batch_fuse, rows_batched, est_wgs = tg_b, rf, est_f(rf)
        env_b = os.environ.get('C1V_WGRAD_BATCH', 'auto')
        self.batch_wgrads = (env_b != '0' and not bn and self.world == 1 and not self.force_bucketed and
                             len(conv_layers) + 2 <= L.WGRAD_BATCH_MAX and os.environ.get('C1V_WGRAD_FUSE', '') == '' and
                             (env_b == '1' or est_wgs <= int(os.environ.get('C1V_WGRAD_BATCH_MAX_WGS', '640'))))
        wgrad_rows = int(os.environ.get('C1V_WGRAD_ROWS', str(rows_batched) if self.batch_wgrads else '384'))

        # Splits of the row (reduction) dimension of a conv wgrad.  Every split is one workgroup per (128 x 128 tile, tap) and
        # one slab for c1v_grad_finalize to sum.  Small batches: as many as keep `wgrad_rows` rows per split.  Large batches
        # (C4: 270 k rows; round 1 capped every layer at 32 splits) -- measured per layer at C4, `C1V_WGRAD_WGS` scans in
        # profiles/r02b_c4_wgrad_splits.txt: what matters is how the wgrad workgroups fill the 2 x 256 workgroup slots of
        # the chip.  In a paired launch they are dispatched first and each runs for the whole launch next to a stream of
        # short dgrad tiles: about 192 of them (under one per CU) is best, more only lengthens the longest CU's queue (a
        # 512-channel layer went from 32 splits = 1536 workgroups to 4 splits: 124 -> 72 us).  A wgrad launched alone (the
        # first convolution) wants both slots of every CU: just under 512 (one workgroup over a full round doubles the time).
        # (round 3, after the wgrad epilogue through LDS and the XCD-aware deal of its workgroups: 320-384 is 0.7 % of the C4 step
        # better than 192 -- alternating runs on one box, 6.573 / 6.581 ms against 6.624 -- and 448+ is worse again)
        wgrad_wgs = int(os.environ.get('C1V_WGRAD_WGS', '320'))
        wgrad_wgs_alone = int(os.environ.get('C1V_WGRAD_WGS_ALONE', '512'))
        wgrad_max = int(os.environ.get('C1V_WGRAD_MAX_SPLITS', '128'))

        def row_splits(rows: int, tile_taps: int, alone: bool) -> int:
            by_rows = rows // wgrad_rows
            by_fill = (wgrad_wgs_alone if alone else wgrad_wgs) // max(1, tile_taps)      # (rounded down)
            return int(max(1, min(by_rows, by_fill, wgrad_max)))

        # Taps fused into one workgroup (c1v_wgrad_args::tap_group; round 3): both operands come into the CU once for all
        # taps, which is what bounded the per-tap form at C4.  Needs 128 x 128 tiles x row splits alone to fill the chip
        # (one 249-register workgroup per CU); such a wgrad is a launch of its own, not paired with the layer's dgrad.
        # OPT-IN (C1V_WGRAD_FUSE=1, or =auto: by size).  Measured at C4 (profiles/r03_c4_wgrad_fused.txt): alone a fused wgrad
        # takes 54-57 us where the per-tap form takes 79-89, and reads each operand from HBM once instead of 2-4 times; but
        # the per-tap form shares its grid with the layer's dgrad tiles (c1v_bwd_pair, two workgroups per CU) and the PAIR
        # finishes in 96-140 us where fused wgrad + dgrad take 106-155 one after the other.
        fuse_env = os.environ.get('C1V_WGRAD_FUSE', '')
        fuse_wgs = int(os.environ.get('C1V_WGRAD_WGS_FUSED', '256'))
        self.wgrad_fused: Dict[str, int] = {}

        def conv_wgrad(X: Buf, dO: Buf, wname: str, bname: str, C_in: int, F: int, flip: int, tap_stride, si, sj, frozen=False,
                       alone=False):
            rows = B * X.pitch
            tiles = (-(-C_in // 128)) * (-(-F // 128))
            tg = 0
            if k > 1:
                tgc = 3 if k % 3 == 0 else (2 if k % 2 == 0 else 0)           # (the group size must divide the taps)
                groups = k // tgc if tgc else 0
                if tgc and (fuse_env == '1' or (fuse_env == 'auto' and tiles * groups * (rows // wgrad_rows) >= 256)):
                    tg = tgc
            if self.batch_wgrads and self.batch_fuse and k > 1:
                tg = self.batch_fuse
                sp = int(max(1, min(rows // wgrad_rows, wgrad_max)))
            elif tg:
                # the workgroups of one row split go to one XCD (32 CUs, one workgroup each): whole splits per XCD
                per_split = tiles * (k // tg)
                sp = 8 * max(1, (fuse_wgs // 8) // per_split)
                sp = int(max(1, min(rows // wgrad_rows, sp, wgrad_max)))
            else:
                sp = row_splits(rows, k * tiles, alone)
            self.wgrad_fused[wname] = tg
            wt = slab(wname, sp, frozen)
            bt = slab(bname, sp, frozen)
            count = int(np.prod(self.seg_shapes[wname]))
            if frozen:
                return None
            return ('c1v_conv1d_wgrad', L.WgradArgs(X.t.data_ptr(), X.cp, C_in, dO.t.data_ptr(), dO.cp, F,
                                                    self.code, k - 1, rows, -(k - 1), k, flip, tap_stride, si, sj,
                                                    wt.data_ptr(), count, sp, bt.data_ptr(), F, tg))

        def dense_wgrad(a_ptr, lda, I, d_ptr, ldd, J, wname, bname, frozen=False):
            wt = slab(wname, 1, frozen)
            bt = slab(bname, 1, frozen)
            if frozen:
                return None
            return ('c1v_dense_wgrad', L.WgradArgs(a_ptr, lda, I, d_ptr, ldd, J, self.code, 0, B, 0, 1, 0, 0, J, 1,
                                                   wt.data_ptr(), I * J, 1, bt.data_ptr(), J))

        # Backward as stages.  A stage = the data gradient of a layer (continues the chain) and the weight gradient of
        # the same layer; both only need the stage's dO buffer.  Default: ONE launch per stage (c1v_bwd_pair, the two
        # GEMMs share a grid).  C1V_BWD_PAIR=0 keeps the older form: dgrad chain on the main stream, wgrads on a side
        # stream (the captured graph keeps both branches; every cross-queue edge of it costs 5-10 us).
        stages = []           # (main ops, side ops, fused ops)
        wg_items = []         # the weight-gradient problems of all stages (c1v_wgrad_batch)

        def stage(dgrad, wgrad, before=(), after=()):
            """dgrad / wgrad: (entry point, args) or None; before / after: extra ops (side / main in the two-stream form)."""
            main = ([op(*dgrad)] if dgrad else []) + list(after)
            side = list(before) + ([op(*wgrad)] if wgrad else [])
            if wgrad:
                wg_items.append(wgrad[1])
            if dgrad and wgrad and getattr(wgrad[1], 'tap_group', 0) >= 2:
                # a fused-tap wgrad fills the chip by itself and runs one 232-register workgroup per CU: own launch
                fused = side + main
            elif dgrad and wgrad:
                is_dense = isinstance(dgrad[1], L.DenseArgs)
                self._keep += [dgrad[1], wgrad[1]]
                pa = L.BwdPairArgs(None if is_dense else C.pointer(dgrad[1]), C.pointer(dgrad[1]) if is_dense else None,
                                   C.pointer(wgrad[1]))
                fused = list(before) + [op('c1v_bwd_pair', pa)] + list(after)
            else:
                fused = side + main
            stages.append((main, side, fused))

        def bn_slabs(name: str, frozen=False):
            """gamma / beta gradients = the two sums of the normalisation's backward reduction (beta first)."""
            st = self.bn[name]
            ch = st['C']
            for w_, o_ in (('/beta', 0), ('/gamma', ch)):
                segs.append(L.SegDesc(st['slab'].data_ptr() + 4 * o_, ch, 1, 1, 0, self.seg_off[name + w_], ch, 1 if frozen else 0, fblk[0]))
                seg_names.append(name + w_)
                fblk[0] += (ch + L.CHUNK - 1) // L.CHUNK

        C_last = g.dec_filters[-1]
        if bn:
            e = self._epi(self.dUd[nd - 1].seq(), L.ROW_STORE)
        else:
            e = self._epi(self.dOd[nd - 1].seq(), L.ROW_PAIRSUM, act=self.Xf.seq(), act_rowmul=2, act_scale=1.0)
        stage(('c1v_conv1d_dgrad', L.ConvArgs(self.dOf.seq(), g.pad_left, g.output_text_size, k,
                                              self.wop['out_dg'].data_ptr(), self.wld['out_dg'], C_last, e)),
              conv_wgrad(self.Xf, self.dOf, 'deconv_decoder_embeddings/kernel', 'deconv_decoder_embeddings/bias',
                         C_last, self.V, 0, C_last * self.V, self.V, 1))
        for i in reversed(range(nd)):
            C_in = g.width if i == 0 else g.dec_filters[i - 1]
            F = g.dec_filters[i]
            nm = 'deconv_decoder_%d_deconv1d_part2' % (i + 1)
            wg = conv_wgrad(self.Xd[i], self.dOd[i], nm + '/kernel', nm + '/bias', C_in, F, 1, F * C_in, 1, C_in)
            pre = []
            if bn:
                bname = 'deconv_decoder_batchnorm_%d' % (i + 1)
                pre = self._bn_bwd_ops(bname, self.Ad[i], L.BN_UP2, self.dUd[i].seq(), self.dOd[i].seq())
                bn_slabs(bname)
            if i > 0 and bn:
                e = self._epi(self.dUd[i - 1].seq(), L.ROW_STORE)
            elif i > 0:
                e = self._epi(self.dOd[i - 1].seq(), L.ROW_PAIRSUM, act=self.Xd[i].seq(), act_rowmul=2, act_scale=1.0)
            else:
                e = self._epi(self.dDense.seq(), L.ROW_STORE, act=self.Xd[0].seq(), act_rowmul=1, act_scale=1.0)
            stage(('c1v_deconv1d_dgrad', L.ConvArgs(self.dOd[i].seq(), k - 1, g.dec_in_len[i], k,
                                                    self.wop['dec%d_dg' % i].data_ptr(), self.wld['dec%d_dg' % i], C_in, e)), wg,
                  before=pre)
        wg = dense_wgrad(self.zd.data_ptr(), self.Dp, self.D, self.dDense.t.data_ptr(), self.nd_p, self.nd,
                         'deconv_decoder_dense/kernel', 'deconv_decoder_dense/bias')
        if not self.frozen and self.D % 4 == 0 and os.environ.get('C1V_LATENT_FUSE', '1') == '1':
            # the gradient of [z_mean | z_log_var] comes straight out of the dense dgrad's epilogue (C1V_ROW_LATENT_BWD):
            # no dzd round trip, no latent_bwd launch
            dmlv_seq = L.Seq(self.dmlv.data_ptr(), self.code, B, self.D2p, 0, 1, self.D2p, 2 * self.D)
            e = self._epi(dmlv_seq, L.ROW_LATENT_BWD)
            e.lat_mlv, e.lat_eps, e.lat_keep = self.mlv.data_ptr(), self.eps.data_ptr(), self.keep_d.data_ptr()
            e.lat_D, e.lat_keep_prob, e.lat_std_scale = self.D, KEEP_PROB, self.std_scale
            e.lat_kc = 1.0 / (self.D * self.Bg)
            stage(('c1v_dense_dgrad', L.DenseArgs(self.dDense.t.data_ptr(), self.code, B, self.nd_p, self.nd_p,
                                                  self.wop['dense_dg'].data_ptr(), self.wld['dense_dg'], self.D, e)), wg)
        elif not self.frozen:
            dzd_seq = L.Seq(self.dzd.data_ptr(), L.F32, B, self.D, 0, 1, self.D, self.D)
            stage(('c1v_dense_dgrad', L.DenseArgs(self.dDense.t.data_ptr(), self.code, B, self.nd_p, self.nd_p,
                                                  self.wop['dense_dg'].data_ptr(), self.wld['dense_dg'], self.D,
                                                  self._epi(dzd_seq))), wg,
                  after=[op('c1v_latent_head_bwd', self._lat_train)])
        else:
            stage(None, wg)
        wg = dense_wgrad(self.H.t.data_ptr(), self.flat_p, self.flat, self.dmlv.data_ptr(), self.D2p, 2 * self.D,
                         'z_heads/kernel', 'z_heads/bias', self.frozen)
        dg = None
        if not self.frozen:
            F = g.n_filters[-1]
            if bn:
                # gradient of the pooled (and dropped-out) output; the normalisation's backward un-pools and applies the mask
                e = self._epi(self.dPe[nl - 1].seq(), L.ROW_STORE, reshape_w=F)
            else:
                e = self._epi(self.dOe[nl - 1].seq(), L.ROW_UNPOOL, reshape_w=F, act=self.H.seq(), act_scale=1.0 / KEEP_PROB,
                              sel=self.sel[nl - 1].data_ptr(), sel_bstride=g.enc_len_pool[nl - 1] * F, sel_cp=F)
            dg = ('c1v_dense_dgrad', L.DenseArgs(self.dmlv.data_ptr(), self.code, B, self.D2p, self.D2p,
                                                 self.wop['heads_dg'].data_ptr(), self.wld['heads_dg'], self.flat, e))
        stage(dg, wg)
        for l in reversed(range(nl)):
            C_in = g.input_vector_size if l == 0 else g.n_filters[l - 1]
            F = g.n_filters[l]
            nm = 'encoder_conv1d_%d' % (l + 1)
            wg = conv_wgrad(self.Xe[l], self.dOe[l], nm + '/kernel', nm + '/bias', C_in, F, 0, C_in * F, F, 1, self.frozen,
                            alone=(l == 0))                 # the first convolution has no data gradient: its wgrad is a launch of its own
            pre = []
            if bn:
                bname = 'encoder_batchnorm_%d' % (l + 1)
                bn_slabs(bname, self.frozen)
                if not self.frozen:
                    last = l == nl - 1
                    pre = self._bn_bwd_ops(bname, self.Ae[l], L.BN_POOL2, self.dPe[l].seq(), self.dOe[l].seq(),
                                           sel=self.sel[l].data_ptr(), sel_bstride=g.enc_len_pool[l] * F, sel_cp=F,
                                           keep=self.keep_e.data_ptr() if last else None)
            dg = None
            if l > 0 and not self.frozen:
                Fp = g.n_filters[l - 1]
                if bn:
                    e = self._epi(self.dPe[l - 1].seq(), L.ROW_STORE)
                else:
                    e = self._epi(self.dOe[l - 1].seq(), L.ROW_UNPOOL, act=self.Xe[l].seq(), act_scale=1.0,
                                  sel=self.sel[l - 1].data_ptr(), sel_bstride=g.enc_len_pool[l - 1] * Fp, sel_cp=Fp)
                dg = ('c1v_conv1d_dgrad', L.ConvArgs(self.dOe[l].seq(), 0, g.enc_in_len[l], k,
                                                     self.wop['enc%d_dg' % l].data_ptr(), self.wld['enc%d_dg' % l], Fp, e))
            stage(dg, wg, before=pre)
        self.bwd_stages = stages
        self.side_stream = torch.cuda.Stream(self.dev)
        # Small batches: the chain of data gradients as launches of their own, then ALL weight gradients in one launch
        # (c1v_wgrad_batch).  At C2 a dgrad alone takes 7-9 us and a paired dgrad + wgrad launch 9-15 (the wgrad workgroups,
        # a few dozen with the longer K loops, are each pair's long pole) and no wgrad feeds the chain: backward 57 -> 46 us.
        # Taken when all of them together are about one round of the chip's 2 x 256 workgroup slots (C1V_WGRAD_BATCH=0 / 1:
        # never / whenever there are at most 12); larger batches keep the pairs, whose dgrad tiles fill the CUs the wgrad
        # workgroups leave idle.  One rank only: the data-parallel step finalizes its first gradient bucket mid-backward.
        self.bwd_batch = None
        self.bwd_overlap = 0
        if self.batch_wgrads and 1 <= len(wg_items) <= L.WGRAD_BATCH_MAX:
            # (longest reductions first: they are the launch's long pole wherever they start)
            order = sorted(wg_items, key=lambda a: -((a.r_end - a.r_begin) // max(1, a.splits)))
            ba = L.WgradBatchArgs(len(order), (C.POINTER(L.WgradArgs) * L.WGRAD_BATCH_MAX)(*[C.pointer(a) for a in order]))
            self._keep += order
            self.bwd_batch = op('c1v_wgrad_batch', ba)
            # Measured and NOT taken (C1V_WGRAD_OVERLAP, default 0): weight gradients on a SECOND stream next to the chain of
            # data gradients, whose launches leave most CUs idle (c1v_run_list_fork).  =1: the first stage's wgrad forks off
            # (it only needs the loss gradient), the rest stays one batch behind the chain; =2: every stage's wgrad forks
            # off behind the dgrad that produces its input.  Bit-identical, but every cross-stream edge costs more than
            # it hides: C2 step 0.185 ms on one stream, 0.203 with one fork + join, 0.221 with four.
            self.bwd_overlap = int(os.environ.get('C1V_WGRAD_OVERLAP', '0'))
            firsts = [st_[1][-1] for st_ in stages if st_[1]]                       # the stages' own wgrad ops, in stage order
            if self.bwd_overlap == 1 and len(order) >= 2 and stages[0][1]:
                rest = [a for a in order if a is not wg_items[0]]
                br = L.WgradBatchArgs(len(rest), (C.POINTER(L.WgradArgs) * L.WGRAD_BATCH_MAX)(*[C.pointer(a) for a in rest]))
                self.bwd_batch_rest = op('c1v_wgrad_batch', br)
            elif self.bwd_overlap == 2 and len(firsts) >= 2:
                pass
            else:
                self.bwd_overlap = 0

        # ---- optimizer
        f_fin = self.lib.c1v_grad_finalize
        f_sq = self.lib.c1v_sqnorm_partial
        f_adv = self.lib.c1v_nadam_advance

        def seg_table(keep):
            """Descriptor table over the segments selected by `keep(name)` (block numbering restarts per table)."""
            sel, nb = [], 0
            for d, name in zip(segs, seg_names):
                if keep(name):
                    sel.append(L.SegDesc(d.slab, d.slab_stride, d.splits, d.reps, d.rep_stride, d.offset, d.count, d.frozen, nb))
                    nb += (d.count + L.CHUNK - 1) // L.CHUNK
            arr = (L.SegDesc * len(sel))(*sel)
            return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev), len(sel), nb

        self.seg_descs, self.n_seg, self.fin_blocks = seg_table(lambda n: True)
        self.sq_partial = torch.zeros(max(self.fin_blocks, 256), dtype=torch.float32, device=self.dev)
        bucketed = self.world > 1 or self.force_bucketed
        n_partial = 256 if bucketed else self.fin_blocks

        f_fin_loss = self.lib.c1v_grad_finalize_loss
        # large batches: the per-sample means in a launch of their own (one wave per sample) instead of inside the single
        # loss workgroup of grad_finalize (C4: that block took 70 us, the rest of the launch 31)
        pre = 1 if B * g.output_text_size >= int(os.environ.get('C1V_LOSS_PRESUM_MIN_ROWS', '65536')) else 0
        loss_args = L.LossReduceArgs(self.xent_rows.data_ptr(), self.ids_in.data_ptr(), self.kl_b.data_ptr(), B,
                                     g.output_text_size, self.Bg, self.loss_out.data_ptr(), self.xent_b.data_ptr(), pre)
        self._keep.append(loss_args)
        presum = [op('c1v_loss_sample_means', loss_args)] if pre else []

        def make_finalize(table, with_sq, with_loss):
            """Sum the wgrad slabs of the table's segments into the flat gradient; with_loss: the step's loss scalars
            ride along as one extra workgroup of the same launch."""
            descs, n, blocks = table
            self._keep.append(descs)

            a = L.GradFinalizeArgs(descs.data_ptr(), n, blocks, self.grads.data_ptr(),
                                   self.sq_partial.data_ptr() if with_sq else None,
                                   C.cast(C.pointer(loss_args), C.c_void_p) if with_loss else None)
            return op('c1v_grad_finalize_step', a)

        finalize = make_finalize((self.seg_descs, self.n_seg, self.fin_blocks), True, True)
        # data-parallel buckets (reverse layer order): the decoder-side gradients are complete after the
        # first `n_dec_stages` backward stages and can be summed over the ranks while the rest of backward runs.  The
        # flat buffer keeps the segments in layer order, so each bucket is one contiguous range of `grads`.
        # C2 with three buckets: decoder side (0.8 MB) after nd+2 stages, the latent heads (3.9 MB, the bulk of the model)
        # after the next stage -- on the wire while the encoder convolutions' backward and its finalize run -- and the
        # encoder convolutions (0.5 MB) at the end; C1V_DP_BUCKETS=2 keeps heads + encoder convolutions in one bucket.
        is_conv = lambda n: n.startswith(('encoder_conv1d_', 'encoder_batchnorm_'))
        is_heads = lambda n: n.startswith('z_heads')
        is_enc = lambda n: is_conv(n) or is_heads(n)
        off_h, off_d = self.seg_off['z_heads/kernel'], self.seg_off['deconv_decoder_dense/kernel']
        assert 0 < off_h < off_d < self.P and all((self.seg_off[n] < off_h) == is_conv(n) and
                                                  (self.seg_off[n] >= off_d) == (not is_enc(n)) for n in self.seg_names)
        # (bucket = backward stages [previous hi, hi), finalize launch over its segments, contiguous slice of `grads`)
        fin_dec = make_finalize(seg_table(lambda n: not is_enc(n)), False, True)
        if presum:
            fin_dec_only = fin_dec

            def fin_dec():
                presum[0]()
                fin_dec_only()
        self.dp_buckets = [(nd + 2, fin_dec, (off_d, self.P))]
        if os.environ.get('C1V_DP_BUCKETS', '3') == '3':
            self.dp_buckets += [(nd + 3, make_finalize(seg_table(is_heads), False, False), (off_h, off_d)),
                                (None, make_finalize(seg_table(is_conv), False, False), (0, off_h))]
        else:
            self.dp_buckets += [(None, make_finalize(seg_table(is_enc), False, False), (0, off_d))]

        def sqnorm():
            L.check(f_sq(self.grads.data_ptr(), self.P, self.sq_partial.data_ptr(), 256, self._stream()), 'c1v_sqnorm_partial')

        def advance():
            L.check(f_adv(self.nadam_state.data_ptr(), self.sched.data_ptr(), self.sq_partial.data_ptr(), n_partial,
                          self.step_ctr.data_ptr(), self._stream()), 'c1v_nadam_advance')

        self.ops_finalize = presum + [finalize]
        nad = L.NadamArgs(self.params.data_ptr(), self.grads.data_ptr(), self.nadam_m.data_ptr(), self.nadam_v.data_ptr(),
                          self.P, 1.0, CLIPNORM, self.sched.data_ptr(), 0.9, 0.999, 1e-7, self.norm_out.data_ptr())
        self.ops_update = ([sqnorm] if bucketed else []) + [advance, op('c1v_nadam_step', nad), self._pack_all]
        # with the one-launch step prologue the schedule scalars are already in `sched` (computed at the start of the step,
        # off the critical path); the Nadam kernel sums the gradient-norm partials itself and moves the RNG counter on
        nad2 = L.NadamArgs(self.params.data_ptr(), self.grads.data_ptr(), self.nadam_m.data_ptr(), self.nadam_v.data_ptr(),
                           self.P, 1.0, CLIPNORM, self.sched.data_ptr(), 0.9, 0.999, 1e-7, self.norm_out.data_ptr(),
                           self.sq_partial.data_ptr(), n_partial, self.sched.data_ptr() + 4 * 7, self.step_ctr.data_ptr())
        # ... and the operand copies of the updated weights are refreshed by the NEXT step's prologue (every fused prologue
        # packs); the inference entry points call _ensure_packed() first
        def mark_stale():
            self._wop_stale = True
        self.ops_update_fused = ([sqnorm] if bucketed else []) + [op('c1v_nadam_step', nad2), mark_stale]

        # ---- RNG (device counter => replayable): eps + both dropout masks in ONE launch; the counter advances in
        # nadam_advance at the end of the step
        thr = int(np.float32(KEEP_PROB) * np.float32(16777216.0))
        reqs = [(self.eps.data_ptr(), B * self.D, 0, 0, 1, self.rank * B * self.D, self.Bg * self.D),
                (self.keep_e.data_ptr(), B * self.flat, 1, thr, 2, self.rank * B * self.flat, self.Bg * self.flat),
                (self.keep_d.data_ptr(), B * self.D, 1, thr, 3, self.rank * B * self.D, self.Bg * self.D)]
        rd, nb = [], 0
        for (ptr, n, kind, th, sid, off, mul) in reqs:
            rd.append(L.RngDesc(ptr, n, kind, th, sid, off, mul, nb, 0))
            nb += (n + 3 + 1023) // 1024 + 1
        arr = (L.RngDesc * len(rd))(*rd)
        self.rng_descs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
        f_m = self.lib.c1v_philox_multi
        f_n = self.lib.c1v_philox_normal
        f_c = self.lib.c1v_counter_add
        n_rd, rng_blocks = len(rd), nb

        def rng():
            L.check(f_m(self.rng_descs.data_ptr(), n_rd, rng_blocks, self.seed, self.step_ctr.data_ptr(), self._stream()), 'philox_multi')

        def rng_eval():
            st = self._stream()
            ctr = self.step_ctr.data_ptr()
            L.check(f_n(self.eps.data_ptr(), B * self.D, self.seed, 4, self.rank * B * self.D, ctr, self.Bg * self.D, st), 'philox')
            L.check(f_c(ctr, 1, st), 'counter_add')
        self.ops_rng = [rng]
        self.ops_rng_eval = [rng_eval]
        self._rng_n, self._rng_blocks = n_rd, rng_blocks

    # ------------------------------------------------------------------------------------------------------
    # execution
    # ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _run(ops):
        for o in ops:
            o()

    def _flat(self, ops):
        """`ops` with the Python-level stage runner replaced by the launches it issues, or None if an op of the list cannot
        be recorded (the side-stream backward, ops that are not `f(args*, stream)` entry points)."""
        out, flags = [], []
        join_next = False

        def put(o, f=0):
            out.append(o)
            flags.append(f)
        for o in ops:
            if getattr(o, '__func__', None) is VAEEngine._run_backward:
                if self.bwd_batch is not None and self.bwd_overlap == 1:
                    put(self.bwd_stages[0][1][-1], L.LIST_SIDE | L.LIST_FORK)
                    for main_ops, _, _ in self.bwd_stages:
                        for m_ in main_ops:
                            put(m_)
                    put(self.bwd_batch_rest)
                    join_next = True
                    continue
                if self.bwd_batch is not None and self.bwd_overlap == 2:
                    with_wg = [i for i, st_ in enumerate(self.bwd_stages) if st_[1]]
                    for i, (main_ops, side_ops, _) in enumerate(self.bwd_stages):
                        if side_ops and i != with_wg[-1]:
                            put(side_ops[-1], L.LIST_SIDE | L.LIST_FORK)
                        for m_ in main_ops:
                            put(m_)
                        if side_ops and i == with_wg[-1]:
                            put(side_ops[-1])
                    join_next = True
                    continue
                if self.bwd_batch is not None:
                    for main_ops, _, _ in self.bwd_stages:
                        for m_ in main_ops:
                            put(m_)
                    put(self.bwd_batch)
                    continue
                if not (os.environ.get('C1V_BWD_PAIR', '1') == '1' or self.g.use_batch_norm):
                    return None
                for _, _, fused_ops in self.bwd_stages:
                    for m_ in fused_ops:
                        put(m_)
            elif getattr(o, '__name__', '') == 'mark_stale':
                continue                                   # (the caller sets the flag, as for graph replays)
            else:
                put(o, L.LIST_JOIN if join_next else 0)
                join_next = False
        if not any(flags):
            out = self._fused_head(out)
            flags = [0] * len(out)
        self._list_blockers = [getattr(o, '__name__', repr(o)) for o in out if not hasattr(o, 'c_fn')]    # (diagnostic)
        return (out, flags) if not self._list_blockers else None

    def _list(self, key: str, ops) -> bool:
        """Issue `ops` back to back from native code (c1v_run_list): one ctypes call for the whole step.  False if the list
        cannot be recorded."""
        rec = self._lists.get(key)
        if rec is None:
            fl = self._flat(ops)
            if fl is None:
                self._lists[key] = rec = False
            else:
                flat, flags = fl
                n = len(flat)
                fns = (C.c_void_p * n)(*[C.cast(o.c_fn, C.c_void_p) for o in flat])
                args = (C.c_void_p * n)(*[C.cast(C.pointer(o.c_args), C.c_void_p) for o in flat])
                fa = (C.c_int32 * n)(*flags) if any(flags) else None
                self._lists[key] = rec = (n, fns, args, flat, fa)
        if rec is False:
            return False
        if rec[4] is not None:
            rc = self.lib.c1v_run_list_fork(rec[0], rec[1], rec[2], rec[4], self._stream(),
                                            C.c_void_p(self.side_stream.cuda_stream))
        else:
            rc = self.lib.c1v_run_list(rec[0], rec[1], rec[2], self._stream())
        if rc != 0:
            L.check(rc, 'c1v_run_list')
        return True

    def _launch(self, key: str, ops):
        """A recorded op list of the inference / validation entry points: from native code if every op can be listed
        (c1v_run_list), else as a HIP graph."""
        if os.environ.get('C1V_STEP_LAUNCH', 'list') == 'list' and self._list(key, ops):
            return
        self._graph(key, ops)

    def _graph(self, key: str, ops):
        """Run `ops` through a HIP graph captured on first use (eager warm-up first so that one-time kernel
        attribute calls happen outside the capture)."""
        gr = self._graphs.get(key)
        if gr is None:
            ops = self._fused_head(ops)
            torch.cuda.synchronize(self.dev)
            saved = self._save_state()
            self._saved_for_capture = saved
            s = torch.cuda.Stream(self.dev)
            s.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(s):
                self._run(ops)
            torch.cuda.current_stream(self.dev).wait_stream(s)
            torch.cuda.synchronize(self.dev)
            self._restore_state(saved)
            gr = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while this thread captures (only matters for graphs
            # that contain collectives; harmless otherwise)
            with torch.cuda.graph(gr, capture_error_mode='thread_local'):
                self._run(ops)
            self._restore_state(saved)
            self._graphs[key] = gr
        gr.replay()

    def _state_tensors(self):
        bn = [st[k_] for st in self.bn.values() for k_ in ('moving_mean', 'moving_var')]
        return [self.params, self.nadam_m, self.nadam_v, self.nadam_state, self.step_ctr] + bn

    def _save_state(self):
        return [t.clone() for t in self._state_tensors()]

    def _restore_state(self, saved):
        for t, s in zip(self._state_tensors(), saved):
            t.copy_(s)
        self._pack_all()

    def set_input_table(self, table: np.ndarray):
        """Device-side batch assembly (SequenceForVAE.__getitem__ :1374-1390 without the Python loops): `table` is the
        input word-vector matrix [N_in, V_in]; a batch is then given as int32 row ids [B, T_in] (-1 = all-zero row) in
        `in_ids` and c1v_gather_input builds the encoder operand directly."""
        t = np.ascontiguousarray(table, dtype=np.float32)
        if t.ndim != 2 or t.shape[1] != self.g.input_vector_size:
            raise ValueError('input table must be [N, {0}]'.format(self.g.input_vector_size))
        # (rows padded to whole 16-byte words: c1v_gather_input then reads a row as float4 words)
        ld = _pad(t.shape[1], 4)
        self.in_table = torch.zeros(t.shape[0], ld, dtype=torch.float32, device=self.dev)
        self.in_table[:, :t.shape[1]].copy_(torch.from_numpy(t))
        self.in_ids = torch.full((self.B, self.g.input_text_size), -1, dtype=torch.int32, device=self.dev)
        self._build_gather()
        self._graphs = {k: v for k, v in self._graphs.items() if not k.endswith('@ids')}
        self._lists = {k: v for k, v in self._lists.items() if not k.endswith('@ids')}

    def _build_gather(self):
        self._gather_args = L.GatherInputArgs(
            self.in_table.data_ptr(), self.in_table.shape[1], self.in_ids.data_ptr(), self.g.input_text_size,
            self.g.input_vector_size, self.Xe[0].seq())
        self.ops_gather_in = [self._op('c1v_gather_input', self._gather_args)]
        self._prologue_cache = {}

    def _in_ops(self, from_ids: bool):
        if not from_ids:
            return self.ops_pack_in
        if self.ops_gather_in is None:
            raise L.C1VError('from_ids=True needs set_input_table() first')
        return self.ops_gather_in

    def load_batch(self, x: Optional[np.ndarray] = None, ids: Optional[np.ndarray] = None):
        if x is not None:
            self.x_in.copy_(torch.as_tensor(x, dtype=torch.float32), non_blocking=True)
        if ids is not None:
            self.ids_in.copy_(torch.as_tensor(ids, dtype=torch.int32), non_blocking=True)

    def _run_backward(self, lo: int = 0, hi: Optional[int] = None):
        """Stages [lo, hi) of backward: one paired dgrad + wgrad launch per stage (default), or the dgrad chain on the
        current stream with the wgrads (and the loss scalars) on the side stream, joined at the end."""
        if self.bwd_batch is not None and lo == 0 and hi is None:
            for main_ops, _, _ in self.bwd_stages:
                self._run(main_ops)
            self.bwd_batch()
            return
        if os.environ.get('C1V_BWD_PAIR', '1') == '1' or self.g.use_batch_norm:
            for _, _, fused_ops in self.bwd_stages[lo:hi]:
                self._run(fused_ops)
            return
        main = torch.cuda.current_stream(self.dev)
        side = self.side_stream
        for main_ops, side_ops, _ in self.bwd_stages[lo:hi]:
            if side_ops:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    self._run(side_ops)
            self._run(main_ops)
        main.wait_stream(side)

    def _prologue_op(self, with_rng: bool, from_ids: bool, advance: bool = False):
        """One launch for the random draws, the encoder operand, the live-row table and (advance: the step ends in an
        optimizer update) the Nadam schedule scalars of a train step (c1v_step_prologue)."""
        key = (with_rng, from_ids, advance)
        if key not in self._prologue_cache:
            if from_ids and self.ops_gather_in is None:
                raise L.C1VError('from_ids=True needs set_input_table() first')
            own = self._input_own_launch()
            a = L.StepPrologueArgs(self.rng_descs.data_ptr() if with_rng else None, self._rng_n if with_rng else 0,
                                   self._rng_blocks if with_rng else 0, self.seed, self.step_ctr.data_ptr(),
                                   None if from_ids or own else C.pointer(self._pack_in_args),
                                   C.pointer(self._gather_args) if from_ids and not own else None,
                                   C.pointer(self._live_args) if self.compact and not own else None,
                                   self.nadam_state.data_ptr() if advance else None, self.sched.data_ptr() if advance else None,
                                   self.pack_descs.data_ptr(), self.n_pack, self.pack_blocks)
            self._prologue_cache[key] = self._op('c1v_step_prologue', a)
        return self._prologue_cache[key]

    def _fused_prologue(self) -> bool:
        return os.environ.get('C1V_PROLOGUE', '1') == '1'

    def _live_gather_op(self):
        """c1v_live_rows + c1v_gather_input in one launch of 1024-thread blocks (large batches)."""
        if getattr(self, '_live_gather_cached', None) is None or self._live_gather_cached[0] is not self._gather_args:
            if self.ops_gather_in is None:
                raise L.C1VError('from_ids=True needs set_input_table() first')
            f = self.lib.c1v_live_rows_gather
            la, ga = self._live_args, self._gather_args

            def run():
                L.check(f(C.byref(la), C.byref(ga), self._stream()), 'c1v_live_rows_gather')
            self._live_gather_cached = (ga, run)
        return self._live_gather_cached[1]

    def _input_own_launch(self) -> bool:
        """A large batch (C4: a 168 MB encoder operand, 262 k target rows) takes the input assembly and the live-row table
        out of the prologue into launches of their own: the prologue kernel runs 256-thread blocks, so its single
        live-row block scans with 4 waves instead of the 16 of `c1v_live_rows` (measured at C4: 191 us against 43 us)
        and the gather runs at the occupancy of the prologue's widest job.  Small batches (C2: 5 MB, 8 k rows) keep both
        inside the prologue: two launches less."""
        Xe = self.Xe[0]
        nbytes = self.B * self.g.input_text_size * Xe.cp * (2 if self.code == L.BF16 else 4)
        return nbytes > int(os.environ.get('C1V_PROLOGUE_INPUT_MAX', str(32 << 20)))

    def _ensure_packed(self):
        """Operand copies of the weights are refreshed lazily after a fused-prologue train step (see ops_update_fused)."""
        if self._wop_stale:
            self._pack_all()

    def update_ops(self):
        """Optimizer update of a step whose forward_ops were built with advance=True."""
        return self.ops_update_fused if self._fused_prologue() else self.ops_update

    def forward_ops(self, with_rng: bool, from_ids: bool = False, advance: bool = False):
        """advance: the step ends in update_ops() (the fused prologue then also prepares the Nadam schedule)."""
        if self._fused_prologue():
            if not self._input_own_launch():
                own_in = []
            elif from_ids and self.compact:
                own_in = [self._live_gather_op()]
            else:
                own_in = self.ops_live_rows + self._in_ops(from_ids)
            return [self._prologue_op(with_rng, from_ids, advance)] + own_in + self.ops_enc_train + self.ops_lat_train + \
                self.ops_dec_train + self.ops_loss_train_core
        return (self.ops_rng if with_rng else []) + self.ops_live_first + self._in_ops(from_ids) + self.ops_enc_train + \
            self.ops_lat_train + self.ops_dec_train + self.ops_loss_train

    def train_ops(self, with_rng: bool, from_ids: bool = False, advance: bool = False):
        return self.forward_ops(with_rng, from_ids, advance) + [self._run_backward] + self.ops_finalize

    def train_step(self, use_graph: bool = True, fixed_rng: bool = False, all_reduce=None, from_ids: bool = False):
        """One fit step on the batch resident in x_in / ids_in: forward, loss, backward, clip, Nadam, operand refresh.
        fixed_rng: eps / keep_e / keep_d were filled by the caller (parity runs).

        world > 1 (data parallel, one process per GPU): `all_reduce(t)` sums a slice of the flat fp32 gradient over the
        ranks (RCCL over xGMI) and may return a work handle with `.wait()` (torch.distributed `async_op=True`).  The step
        is one captured graph (or, if the collectives cannot be captured, one graph per bucket) over `dp_buckets`:
            A1 = forward + loss + decoder-side backward + finalize(decoder bucket)   -> all-reduce(decoder bucket) starts
            A2 = latent-heads backward + finalize(heads bucket)                      -> runs WHILE bucket 1 is on the wire
            A3 = encoder-convolution backward + finalize(last bucket)                -> runs WHILE bucket 2 is on the wire
            B  = |g|^2 of the reduced gradient, clip, Nadam, operand refresh         -> after all buckets
        Every loss term already carries 1/B_global, so the sum IS the full-batch gradient."""
        if self.N == 0:
            raise L.C1VError('train_step needs the vocabulary matrix (set_vocab)')
        self._dp_all_reduce = all_reduce
        sfx = '@ids' if from_ids else ''

        # How a captured step is issued.  `list` (default for one rank): the recorded launches back to back from native code
        # (c1v_run_list) -- measured at C2 0.196-0.198 ms against 0.204 ms for the replay of the same launches as a HIP
        # graph, whose replays the runtime keeps ~8 us apart; `graph`: the HIP graph (always for the data-parallel step,
        # whose collectives are captured with it).
        as_list = use_graph and os.environ.get('C1V_STEP_LAUNCH', 'list') == 'list'

        def go(key, ops):
            if use_graph:
                k_ = key + ('_fixed' if fixed_rng else '') + sfx
                if not (as_list and self.world == 1 and not self.force_bucketed and self._list(k_, ops)):
                    self._graph(k_, ops)
            else:
                self._run(ops)

        # (graph replays do not run the Python-side `mark_stale` op of update_ops(): set the flag here)
        stale_after = self._fused_prologue()
        if self.world == 1 and not self.force_bucketed:
            go('train', self.train_ops(not fixed_rng, from_ids, True) + self.update_ops())
            self._wop_stale = stale_after
            return
        if all_reduce is None:
            raise L.C1VError('world_size > 1 needs an all_reduce callable')
        def bucket_ops(i):
            """Backward stages and finalize launch of bucket i."""
            lo = self.dp_buckets[i - 1][0] if i else 0
            hi, fin, _ = self.dp_buckets[i]
            return [lambda: self._run_backward(lo, hi), fin]

        def reduce_bucket(i):
            lo, hi = self.dp_buckets[i][2]
            return all_reduce(self.grads[lo:hi])

        nb = len(self.dp_buckets)
        if use_graph and os.environ.get('C1V_DP_GRAPH', '1') == '1' and not getattr(self, '_dp_graph_failed', False):
            # the collectives are captured INTO the step's graph (RCCL supports hipGraph capture): one launch per step,
            # every bucket on the wire while the backward stages of the next one run.  Rehearsed here with one rank only
            # (`bench.py --force-dp-path`), so a capture failure falls back to the graph-per-bucket sequence below.
            waits = []

            def join():
                for w in waits:
                    if w is not None and hasattr(w, 'wait'):
                        w.wait()
                del waits[:]
            ops = self.forward_ops(not fixed_rng, from_ids, True)
            for i in range(nb):
                ops = ops + bucket_ops(i) + [lambda i=i: waits.append(reduce_bucket(i))]
            try:
                go('train_dp1', ops + [join] + self.update_ops())
                self._wop_stale = stale_after
                return
            except Exception as exc:                                   # noqa: BLE001 - capture of a collective refused
                import sys
                print('c1dvae: capturing the RCCL collectives into the step graph failed (%s); using one graph per '
                      'bucket' % (exc,), file=sys.stderr)
                self._dp_graph_failed = True
                self._graphs.pop('train_dp1' + ('_fixed' if fixed_rng else '') + sfx, None)
                if getattr(self, '_saved_for_capture', None) is not None:
                    self._restore_state(self._saved_for_capture)
        works = []
        for i in range(nb):
            go('train_a%d' % (i + 1), (self.forward_ops(not fixed_rng, from_ids, True) if i == 0 else []) + bucket_ops(i))
            works.append(reduce_bucket(i))
        for w in works:
            if w is not None and hasattr(w, 'wait'):
                w.wait()
        go('train_b', self.update_ops())
        self._wop_stale = stale_after

    def eval_step(self, use_graph: bool = True, fixed_rng: bool = False, from_ids: bool = False):
        """Validation loss of the resident batch (dropout off, eps still sampled -- Keras test phase of FullVAE)."""
        self._ensure_packed()
        ops = ([] if fixed_rng else self.ops_rng_eval) + self.ops_live_first + self._in_ops(from_ids) + self.ops_enc_infer + \
            self.ops_lat_eval + self.ops_dec_eval + self.ops_loss_eval
        if use_graph:
            self._launch(('eval_fixed' if fixed_rng else 'eval') + ('@ids' if from_ids else ''), ops)
        else:
            self._run(ops)

    def sample_reconstruct(self, use_graph: bool = True, from_ids: bool = False, postprocess: bool = True) -> torch.Tensor:
        """FullVAE.predict_on_batch (:1462-1464): dropout off, z SAMPLED (the stage-2 generator trains on these), then
        optionally postprocess_reconstructed_word_vectors."""
        self._ensure_packed()
        ops = self.ops_rng_eval + self._in_ops(from_ids) + self.ops_enc_infer + self.ops_lat_eval + self.ops_dec_infer + \
            (self.ops_post if postprocess else [])
        if use_graph:
            self._launch('sample_recon' + ('_post' if postprocess else '') + ('@ids' if from_ids else ''), ops)
        else:
            self._run(ops)
        return self.post if postprocess else self.y_pred()

    def loss(self) -> float:
        return float(self.loss_out[0].item())

    def encode(self, use_graph: bool = True) -> torch.Tensor:
        """vae_encoder_.predict_on_batch (:404): z_mean of the resident batch, [B, latent] fp32 (device)."""
        self._ensure_packed()
        ops = self.ops_pack_in + self.ops_enc_infer + (self.ops_lat_infer if self.heads_splits > 1 else [])   # sums the slabs
        if use_graph:
            self._launch('encode', ops)
        else:
            self._run(ops)
        return self.mlv[:, :self.D]

    def reconstruct(self, use_graph: bool = True, postprocess: bool = True) -> torch.Tensor:
        """vae_decoder_(vae_encoder_(x)) (:440) and, optionally, postprocess_reconstructed_word_vectors (:439)."""
        self._ensure_packed()
        ops = self.ops_pack_in + self.ops_enc_infer + self.ops_lat_infer + self.ops_dec_infer + (self.ops_post if postprocess else [])
        if use_graph:
            self._launch('recon_post' if postprocess else 'recon', ops)
        else:
            self._run(ops)
        return self.post if postprocess else self._y_view()

    def decode_latent(self, z: torch.Tensor) -> torch.Tensor:
        """vae_decoder_.predict_on_batch(z): z [B, latent] fp32 -> y [B, T_out, V] fp32."""
        self._ensure_packed()               # a train step leaves the refresh of the operand copies to the next prologue
        self.mlv_parts.zero_()
        self.mlv[:, :self.D].copy_(z)
        self._run(self.ops_lat_infer + self.ops_dec_infer)
        return self._y_view()

    def y_pred(self) -> torch.Tensor:
        return self._y_view()

    def _y_view(self) -> torch.Tensor:
        """The decoder output [B, T_out, V] fp32 (a strided view: rows lie `Vp` values apart)."""
        To = self.g.output_text_size
        return self.Y.t[: self.B * To * self.Vp].view(self.B, To, self.Vp)[:, :, :self.V]


class VocabDecoder:
    """a10-ii: nearest vocabulary row of each query (gensim `similar_by_vector` arithmetic, :587), optionally with the
    vocabulary sharded over ranks (each rank owns rows [n_base, n_base + N_local))."""

    def __init__(self, vocab_norm: np.ndarray, max_queries: int, dtype: str = 'bf16', device: str = 'cuda:0', n_base: int = 0):
        if not torch.cuda.is_available():
            raise L.C1VError('VocabDecoder needs a GPU')
        self.lib = L.load()
        self.dev = torch.device(device)
        self.tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
        self.code = L.BF16 if dtype == 'bf16' else L.F32
        W = np.asarray(vocab_norm, dtype=np.float32)
        self.N, self.K = W.shape
        self.kp = 32 if self.K <= 32 else _pad(self.K, 64)
        self.n_pad = _pad(self.N, 64)
        self.n_base = int(n_base)
        host = np.zeros((self.n_pad + L.SLACK_ROWS, self.kp), dtype=np.float32)
        host[:self.N, :self.K] = W
        self.w = torch.from_numpy(host).to(self.dev).to(self.tdt).contiguous()
        self.Mmax = int(max_queries)
        self.m_pad = _pad(self.Mmax, 128)
        self.q = torch.zeros((self.m_pad + L.SLACK_ROWS) * self.kp, dtype=self.tdt, device=self.dev)
        fused = self.code == L.BF16 and L.fused_width(self.kp)
        if self.code == L.BF16 and not fused:
            _warn_unfused(self.kp)
        tiles = self.n_pad // 64
        self.splits = max(1, min(tiles, (2 * 256) // max(1, self.m_pad // 128), 16)) if fused else 1
        self.part_score = torch.zeros(self.splits * self.m_pad, dtype=torch.float32, device=self.dev)
        self.part_idx = torch.zeros(self.splits * self.m_pad, dtype=torch.int32, device=self.dev)
        self.score = torch.zeros(self.m_pad, dtype=torch.float32, device=self.dev)
        self.idx = torch.zeros(self.m_pad, dtype=torch.int32, device=self.dev)
        if fused:
            self.n_chunk, self.logits = 0, None
        else:
            self.n_chunk = int(min(self.n_pad, max(1024, (1 << 28) // self.m_pad // 64 * 64)))
            self.logits = torch.zeros(self.m_pad * self.n_chunk, dtype=torch.float32, device=self.dev)

    def load_queries(self, q: torch.Tensor):
        """q [M, K] (any float dtype, device or host): rows are used as given (normalise beforehand)."""
        M = q.shape[0]
        if M > self.Mmax or q.shape[1] != self.K:
            raise ValueError('queries must be [<= {0}, {1}]'.format(self.Mmax, self.K))
        qv = self.q[: self.m_pad * self.kp].view(self.m_pad, self.kp)
        qv[:M, :self.K].copy_(q.to(self.dev))
        self.M = M

    def load_rows_normalized(self, rows: torch.Tensor, ld: int):
        """Queries straight from device rows (e.g. the post-processed decoder output [M, ld] fp32): the first K features of
        every row, L2-normalised by c1v_l2norm_rows (find_best_words :584-587), land in the operand buffer.  No host copy."""
        M = rows.numel() // ld
        if M > self.Mmax or ld < self.K:
            raise ValueError('rows must be [<= {0}, >= {1}]'.format(self.Mmax, self.K))
        a = L.L2NormArgs(rows.data_ptr(), M, self.K, ld, self.q.data_ptr(), self.code, self.kp, None, 1.0)
        L.call('c1v_l2norm_rows', C.byref(a), C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        self.M = M

    def run(self):
        a = L.VocabArgmaxArgs(self.q.data_ptr(), self.code, self.M, self.kp, self.w.data_ptr(), self.N, self.n_pad,
                              self.n_base, self.part_score.data_ptr(), self.part_idx.data_ptr(), self.m_pad, self.splits,
                              self.score.data_ptr(), self.idx.data_ptr(),
                              self.logits.data_ptr() if self.logits is not None else None, self.n_chunk)
        L.call('c1v_vocab_argmax', C.byref(a), C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        return self.score[:self.M], self.idx[:self.M]


class VocabTopN:
    """a10-ii with topn > 1: gensim `similar_by_vector(v, topn)` (reference :587) over a table of unit rows.
    fp32 MFMA GEMM into a logits scratch + fused row top-n (c1v_vocab_topn); queries are processed in row chunks so
    that the scratch stays below `scratch_bytes`."""

    def __init__(self, vocab_norm: np.ndarray, device: str = 'cuda:0', dtype: str = 'f32', scratch_bytes: int = 1 << 28):
        if not torch.cuda.is_available():
            raise L.C1VError('VocabTopN needs a GPU: nearest-word search runs on the HIP kernels only.')
        self.lib = L.load()
        self.device_str = device
        self.dev = torch.device(device)
        self.tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
        self.code = L.BF16 if dtype == 'bf16' else L.F32
        W = np.asarray(vocab_norm, dtype=np.float32)
        self.N, self.K = W.shape
        self.kp = _pad(self.K, 64)
        host = np.zeros((self.N + L.SLACK_ROWS, self.kp), dtype=np.float32)
        host[:self.N, :self.K] = W
        self.w = torch.from_numpy(host).to(self.dev).to(self.tdt).contiguous()
        self.ld = _pad(self.N, 64)
        self.m_chunk = int(max(1, min(4096, scratch_bytes // (4 * self.ld))))
        self.logits = torch.zeros(self.m_chunk * self.ld, dtype=torch.float32, device=self.dev)
        self.q = torch.zeros((self.m_chunk + L.SLACK_ROWS) * self.kp, dtype=self.tdt, device=self.dev)

    def run(self, queries: np.ndarray, topn: int, n_base: int = 0):
        """queries [M, K] (used as given: normalise beforehand).  Returns (score [M, topn] f32, index [M, topn] i32) as
        NumPy arrays, best first, ties -> lowest index."""
        q = torch.as_tensor(np.asarray(queries, dtype=np.float32))
        M = q.shape[0]
        if q.shape[1] != self.K:
            raise ValueError('queries must be [M, {0}]'.format(self.K))
        if not (0 < topn <= self.N):
            raise ValueError('topn must be in [1, {0}]'.format(self.N))
        score = torch.empty((M, topn), dtype=torch.float32, device=self.dev)
        idx = torch.empty((M, topn), dtype=torch.int32, device=self.dev)
        st = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        qv = self.q[: self.m_chunk * self.kp].view(self.m_chunk, self.kp)
        for m0 in range(0, M, self.m_chunk):
            m = min(self.m_chunk, M - m0)
            qv[:m, :self.K].copy_(q[m0:m0 + m].to(self.dev))
            a = L.VocabTopnArgs(self.q.data_ptr(), self.code, m, self.kp, self.w.data_ptr(), self.N, n_base,
                                self.logits.data_ptr(), self.ld, topn, score[m0:].data_ptr(), idx[m0:].data_ptr())
            L.call('c1v_vocab_topn', C.byref(a), st)
        return score.cpu().numpy(), idx.cpu().numpy()


def postprocess_word_vectors(word_vectors: np.ndarray, device: str = 'cuda:0'):
    """a10-i on NumPy data (the static helper `postprocess_reconstructed_word_vectors`, reference :542-565): upload,
    c1v_eos_truncate_normalize (float32) or its float64 variant (any other dtype is computed in float64 and cast back,
    so the dtype of the argument is kept, as the reference does), download.  Returns (out [B,T,V-1], end_idx [B])."""
    if not torch.cuda.is_available():
        raise L.C1VError('postprocess_reconstructed_word_vectors needs a GPU: the HIP kernel is the only implementation.')
    lib = L.load()
    a = np.asarray(word_vectors)
    B, T, V = a.shape
    dev = torch.device(device)
    f32 = a.dtype == np.float32
    src = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32 if f32 else np.float64)).to(dev)
    out = torch.empty((B, T, V - 1), dtype=src.dtype, device=dev)
    end = torch.empty((B,), dtype=torch.int32, device=dev)
    fn = lib.c1v_eos_truncate_normalize if f32 else lib.c1v_eos_truncate_normalize_f64
    L.check(fn(src.data_ptr(), B, T, V, out.data_ptr(), end.data_ptr(),
               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'c1v_eos_truncate_normalize')
    return out.cpu().numpy().astype(a.dtype, copy=False), end.cpu().numpy()
