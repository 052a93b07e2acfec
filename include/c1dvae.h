/* libc1dvae -- C ABI of the MI355X-native Conv1D text-VAE hot path.
 *
 * The reference (nsu-ai-team/conv1d-text-vae) is pure Python on Keras/TensorFlow and has no FFI of its own
 * (SURVEY.md section 8b).  Each entry point below replaces the Keras layer / optimizer call the reference
 * makes at the cited line of conv1d_text_vae/conv1d_text_vae.py; the Python host in conv1d_text_vae_amd/
 * binds them with ctypes.
 *
 * Conventions
 *   - every op is `int c1v_<op>(const c1v_<op>_args*, void* hipStream)`; 0 = ok, negative = error and
 *     `c1v_last_error()` (thread-local) says why.  No exceptions cross the ABI.
 *   - the library never allocates device memory: every pointer is caller-owned device memory
 *     (the Python host allocates with torch).  All launches go to the stream given; nothing synchronises,
 *     so every op can be captured into a HIP graph.
 *   - dtype: 0 = float32, 1 = bfloat16.  "T" below is the compute/storage dtype of a model instance.
 *   - shape checks happen on the host before any launch; a failed check launches nothing.
 */
#ifndef C1DVAE_H
#define C1DVAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C1V_F32 0
#define C1V_BF16 1

/* Channels-last padded sequence buffer.  Element (b, t, c) lives at
 *     data[b * bstride + (off + t) * cp + c]          (element units of `dtype`)
 * Rows outside [off, off+len) of a sample and channels >= c are kept ZERO by every producer; the
 * allocation carries `C1V_SLACK_ROWS` zero rows after the last sample so that window reads and the unpredicated
 * tile loads of the GEMM kernels (up to one 64-row tile) may run over.  No load in the GEMM kernels is predicated:
 * K tails rely on the zero padding of the weight operand rows, row tails on these zero rows. */
typedef struct c1v_seq {
    void*   data;
    int32_t dtype;
    int32_t batch;
    int64_t bstride;   /* elements between samples (= pitch * cp for a conv operand buffer) */
    int32_t off;       /* first valid row inside a sample */
    int32_t len;       /* valid rows */
    int32_t cp;        /* row pitch in elements (multiple of 32 for a conv/GEMM operand) */
    int32_t c;         /* valid channels */
} c1v_seq;

#define C1V_SLACK_ROWS 72

/* row modes of the fused epilogue */
#define C1V_ROW_STORE   0   /* dst row = u                                                    */
#define C1V_ROW_POOL2   1   /* MaxPool1D(2): rows (2i,2i+1) -> i, writes `sel`        (a2)     */
#define C1V_ROW_UP2     2   /* UpSampling1D(2)+Cropping1D: row u -> 2u, 2u+1 (< dst.len) (a6)  */
#define C1V_ROW_PAIRSUM 3   /* backward of UP2: rows (2i,2i+1) summed -> i                     */
#define C1V_ROW_UNPOOL  4   /* backward of POOL2: row u -> 2u+sel (other row <- 0)             */
#define C1V_ROW_LATENT_BWD 5 /* dense dgrad straight into the gradient of [z_mean | z_log_var]: the GEMM value (b, d) is
                                d loss / d zd; the epilogue applies c1v_latent_head_bwd's arithmetic (dropout mask, sampling
                                and KL terms, see the lat_* fields) and stores dst(b, d) and dst(b, lat_D + d).  Vector
                                epilogue only (lat_D a multiple of 4, 16-byte aligned buffers). */
#define C1V_ROW_NORM    6   /* STORE + the first step of the loss in the same pass (a7 + a8, reference :1279-1280 + :1195): the
                                whole output row (n_out <= 320 channels) lies in ONE workgroup, so besides dst(b, u, :) = v the
                                epilogue writes yhat[pos][:] = v / sqrt(max(sum v^2, 1e-12)) * nrm_qscale (compute dtype, zero
                                padded to nrm_kp) and rnorm[row] -- what c1v_l2norm_rows would compute from dst in a launch of its
                                own.  row = b * n_rows + u; pos = nrm_pos ? nrm_pos[row] : row, rows with pos < 0 get no yhat
                                (and, with nrm_live_only, no dst row either: a training step never reads them).  Convolutions
                                over whole 128-byte channel strips only (c1v_conv1d_fwd checks). */

/* Generic fused epilogue applied by conv / dense kernels to each accumulator element (u, n) of sample b:
 *   v += bias[n]; v = relu(v) if relu;
 *   column split: if reshape_w > 0 then (t, c) = divmod(n, reshape_w) else (t, c) = (u, n)
 *   row mode (above) maps t -> destination row(s);
 *   v *= (act(b, row, c) > 0) * act_scale    if act.data   (ReLU / dropout backward mask)
 *   v  = keep[b, row, c] ? v * keep_scale : 0  if keep      (Dropout forward, :1243)
 */
typedef struct c1v_epilogue {
    const float*   bias;        /* [n] or NULL */
    int32_t        relu;
    int32_t        row_mode;
    int32_t        reshape_w;   /* 0 = off */
    c1v_seq        dst;         /* destination; dst.dtype may differ from the compute dtype */
    c1v_seq        act;         /* optional mask source (act.data == NULL: none) */
    int32_t        act_rowmul;  /* mask row = act.off + row * act_rowmul (2 when the activation is stored upsampled) */
    float          act_scale;
    uint8_t*       sel;         /* POOL2: written; UNPOOL: read.  index b*sel_bstride + row*sel_cp + c */
    int64_t        sel_bstride;
    int32_t        sel_cp;
    const uint8_t* keep;        /* dropout keep mask, same indexing as sel */
    float          keep_scale;
    /* C1V_ROW_LATENT_BWD only: mlv [B][2*lat_D] f32, eps [B][lat_D] f32, keep mask [B][lat_D] or NULL, and the scalars of
       c1v_latent_args (lat_kc = 1 / (lat_D * B_global)) */
    const float*   lat_mlv; const float* lat_eps; const uint8_t* lat_keep;
    int32_t        lat_D; float lat_keep_prob, lat_std_scale, lat_kc;
    int32_t        bias_row_w;  /* with reshape_w: the GEMM was laid out with every (t, .) row padded from bias_row_w up to
                                   reshape_w channels (zero weight rows) so that the vector epilogue applies to widths that
                                   are not multiples of 4; bias index = t*bias_row_w + c, pad channels get 0.  0 = no padding */
    /* C1V_ROW_NORM only (see above; the arguments of c1v_l2norm_args) */
    void*          nrm_yhat; int32_t nrm_kp; float* nrm_rnorm; float nrm_qscale; const int32_t* nrm_pos; int32_t nrm_live_only;
    c1v_seq        nrm_dy;      /* optional (data != NULL, with nrm_pos): rows without a target get an all-zero row (b, u) of this
                                   buffer -- the gradient of the loss w.r.t. y, which c1v_cos_softmax_finish then writes for the
                                   live rows only (c1v_xent_finish_args::live_rows) */
} c1v_epilogue;

/* ---- a1 / a6 / a7: Conv1D('valid'|'same') and Conv2DTranspose((k,1)) forward (:1227-1280) -------------
 * out(b,u,n) = sum_{j<k, c<x.cp} X[b][a_off + u + j][c] * w_op[n][j*x.cp + c]   for u < n_rows
 * X rows are buffer rows (x.off is NOT added: zero border rows implement 'same'/'full' padding, so
 * valid conv: a_off=0; same conv: x.off=(k-1)/2; transposed conv: x.off=k-1 with w_op taps flipped).
 * w_op [n_out][k*x.cp] in the compute dtype comes from c1v_pack_weights.  The same entry point computes
 * dgrad (X = padded output-gradient buffer, w_op = dgrad operand copy). */
typedef struct c1v_conv1d_args {
    c1v_seq      x;
    int32_t      a_off;
    int32_t      n_rows;      /* output rows per sample */
    int32_t      k;
    const void*  w_op;
    int64_t      ldw;         /* row stride of w_op in elements: >= k*x.cp, rows zero-padded to whole 128-byte K tiles */
    int32_t      n_out;       /* valid output channels N */
    c1v_epilogue epi;
} c1v_conv1d_args;
int c1v_conv1d_fwd(const c1v_conv1d_args*, void* stream);
int c1v_conv1d_dgrad(const c1v_conv1d_args*, void* stream);     /* same kernel family, named per SURVEY 8b */
int c1v_deconv1d_fwd(const c1v_conv1d_args*, void* stream);     /* requires x.off == k-1 */
int c1v_deconv1d_dgrad(const c1v_conv1d_args*, void* stream);

/* ---- a3 / a5: Dense forward / dgrad (:1244-1245, :1253) -------------------------------------------------
 * out(b,n) = sum_k A[b*lda + k] * w_op[n][k], k < K (K multiple of 8 elements, rows 16-byte aligned; both operands
 * readable up to the next multiple of 128 bytes of K, w_op zero there; A readable for 64 rows past `rows` in wgrad). */
typedef struct c1v_dense_args {
    const void*  a;
    int32_t      dtype;
    int32_t      rows;        /* M = batch */
    int64_t      lda;
    int32_t      K;
    const void*  w_op;        /* [n_out][ldw] */
    int64_t      ldw;
    int32_t      n_out;
    c1v_epilogue epi;
    int32_t      k_splits;    /* 0 / 1 = off.  > 1: K is cut into k_splits ranges of whole 128-byte tiles, one workgroup each
                                 (a GEMM with few output tiles and a long K then covers the chip); split s adds its partial
                                 sums into epi.dst + s*split_stride elements (bias in split 0 only), the consumer sums the
                                 slabs (c1v_latent_head_fwd: mlv_parts).  STORE epilogue without relu / act / keep only. */
    int64_t      split_stride;
} c1v_dense_args;
int c1v_dense_fwd(const c1v_dense_args*, void* stream);
int c1v_dense_dgrad(const c1v_dense_args*, void* stream);

/* ---- wgrad of conv / deconv / dense: dW partial sums (TN GEMM, split over rows) ---------------------------
 * for tap in [0,taps): out[s][tap_off(tap) + i*si + j*sj] = sum_{r in split s} A[(r + a_shift0 + tap)*lda + i] * D[r*ldd + j]
 * with tap_off(tap) = (flip ? taps-1-tap : tap) * tap_stride, i < I, j < J.  The caller sums the `splits`
 * slabs with c1v_grad_finalize.  Rows r run over [r_begin, r_end) of the flat buffers (zero border rows of D
 * make cross-sample terms vanish; A and D must share the per-sample pitch). */
typedef struct c1v_wgrad_args {
    const void* a;  int64_t lda;  int32_t I;
    const void* d;  int64_t ldd;  int32_t J;
    int32_t     dtype;
    int64_t     r_begin, r_end;
    int32_t     a_shift0;
    int32_t     taps;
    int32_t     flip;
    int64_t     tap_stride, si, sj;
    float*      out;
    int64_t     slab_stride;
    int32_t     splits;
    float*      bias_out;        /* optional: bias_out[s*bias_slab_stride + j] = sum_{r in split s} D[r*ldd + j] (bias gradient) */
    int64_t     bias_slab_stride;
    int32_t     tap_group;       /* 0 / 1: one workgroup per (tile, tap, split).  2 / 3: that many consecutive taps share a
                                    workgroup, which fetches both operands once for all of them (a third of the bytes into the
                                    CU per MFMA, 192 accumulator registers per wave): for launches whose tiles x splits fill
                                    the chip without the tap dimension; must divide `taps`.  Results are bit-identical either way. */
} c1v_wgrad_args;
int c1v_wgrad(const c1v_wgrad_args*, void* stream);
int c1v_conv1d_wgrad(const c1v_wgrad_args*, void* stream);
int c1v_deconv1d_wgrad(const c1v_wgrad_args*, void* stream);
int c1v_dense_wgrad(const c1v_wgrad_args*, void* stream);

/* One backward stage in ONE launch: the data gradient (exactly one of conv_dgrad / dense_dgrad; conv_dgrad also
 * serves the transposed conv) and the weight gradient of the same layer.  Both read the stage's output-gradient
 * buffer and neither fills the chip alone, so their tiles share one grid; results are bit-identical to the two
 * separate calls.  This is how the engine runs the backward of model_fn (reference :1283-1329, Keras' autodiff of
 * the a1-a7 layers); the separate entry points above remain for callers that need one of the two. */
typedef struct c1v_bwd_pair_args {
    const c1v_conv1d_args* conv_dgrad;
    const c1v_dense_args*  dense_dgrad;
    const c1v_wgrad_args*  wgrad;
} c1v_bwd_pair_args;
int c1v_bwd_pair(const c1v_bwd_pair_args*, void* stream);

/* column sums (bias gradients): out[s][c] = sum_{rows of split s} buf[row*ld + c], c < C */
typedef struct c1v_colsum_args {
    const void* buf; int32_t dtype; int64_t rows; int64_t ld; int32_t C;
    float* out; int64_t slab_stride; int32_t splits;
} c1v_colsum_args;
int c1v_colsum(const c1v_colsum_args*, void* stream);

/* ---- operand copies of the fp32 master weights ------------------------------------------------------------
 * dst[n*ldk + tap'*cp + c] = src[tap*s_tap + c*s_c + n*s_n],  tap' = flip ? taps-1-tap : tap,
 * for n < n_count, tap < taps, c < c_count.  Padding entries of dst are never written (stay zero). */
#define C1V_PACK_TILE 64   /* c1v_pack_weights: one workgroup per C1V_PACK_TILE x C1V_PACK_TILE tile of one tap */
typedef struct c1v_pack_desc {
    const float* src; void* dst; int32_t dtype;
    int32_t n_count, taps, c_count, cp, flip;
    int64_t ldk, s_tap, s_c, s_n;
    int32_t first_block, pad;   /* descriptor i owns blocks [first_block_i, first_block_{i+1}): taps * ceil(c_count/C1V_PACK_TILE) * ceil(n_count/C1V_PACK_TILE)
                                 * each (one block = one 32x32 tile of one tap, transposed through LDS) */
} c1v_pack_desc;
#define C1V_CHUNK 2048          /* elements handled by one block of the descriptor-table kernels */
/* `descs` is a DEVICE array of n_desc descriptors (ascending first_block); total_blocks = sum of their block counts */
int c1v_pack_weights(const c1v_pack_desc* descs, int32_t n_desc, int32_t total_blocks, void* stream);

/* input featuriser output -> operand buffer: dst(b,t,c) = src[(b*T + t)*C + c] (float32 -> T, zero padded) */
typedef struct c1v_pack_input_args { const float* src; int32_t T, C; c1v_seq dst; } c1v_pack_input_args;
int c1v_pack_input(const c1v_pack_input_args*, void* stream);
/* gather variant (device-side batch assembly, SequenceForVAE :1374-1390): dst(b,t,:) = table[ids[b*T+t]] or 0 if id<0 */
typedef struct c1v_gather_input_args { const float* table; int64_t ld; const int32_t* ids; int32_t T, C; c1v_seq dst; } c1v_gather_input_args;
int c1v_gather_input(const c1v_gather_input_args*, void* stream);

/* ---- BatchNormalization (use_batch_norm=True: :1229-1230, :1238-1239, :1259-1260, :1270-1272) ----------------------------
 * Keras 2.2 defaults: axis -1, momentum 0.99, epsilon 1e-3.  The layer normalises a convolution's post-ReLU output x
 * [B][len][C] per channel and is followed by MaxPool1D(2) (encoder: C1V_BN_POOL2) or UpSampling1D(2) + Cropping1D
 * (decoder: C1V_BN_UP2), which the kernels apply themselves.
 *   c1v_bn_reduce(backward = 0): sums[0][c] = sum x, sums[1][c] = sum x^2 over (b, t)        (times sum_scale)
 *   c1v_bn_reduce(backward = 1): sums[0][c] = sum dy, sums[1][c] = sum dy * xhat, dy = gradient w.r.t. the normalised
 *       output, obtained on the fly from `grad` (POOL2: gradient of the pooled output + sel (+ keep mask); UP2: gradient
 *       of the upsampled, cropped output, rows 2t and 2t+1 summed)
 *     two deterministic stages through `partial` ([blocks][2][C] floats).  Data-parallel ranks all-reduce `sums`.
 *   c1v_bn_stats: mean, var (biased) from the sums of the GLOBAL batch (n_global rows); moving averages, if given:
 *       moving = moving * momentum + (1 - momentum) * {mean, var * n / (n - (1 + eps))}
 *   c1v_bn_apply_fwd: y = gamma * (x - mean) / sqrt(var + eps) + beta -> pool (writes sel_out, applies the keep mask) or
 *       upsample into dst.  Inference passes the moving averages as mean / var.
 *   c1v_bn_apply_bwd: dst(b, t, c) = [x > 0] * gamma / sqrt(var + eps) * (dy - sums[0] / n - xhat * sums[1] / n)
 *       (the gradient at the convolution's pre-activation); gamma's gradient is sums[1], beta's is sums[0]. */
#define C1V_BN_POOL2 1
#define C1V_BN_UP2   2
typedef struct c1v_bn_args {
    c1v_seq x;                 /* post-ReLU conv output; x.len = rows per sample, x.c = channels */
    int32_t mode;
    c1v_seq grad;              /* backward only */
    const uint8_t* sel;        /* POOL2 backward: which row of each pair was kept */
    int64_t sel_bstride; int32_t sel_cp;     /* indexing of sel / sel_out / keep: b*sel_bstride + t2*sel_cp + c */
    const uint8_t* keep; float keep_scale;   /* optional dropout keep mask on the POOLED output (forward and backward) */
    const float* mean; const float* var; float eps;
    const float* gamma; const float* beta;
    const float* sums;         /* backward apply: [2][C] over the global batch */
    int64_t n_global;          /* rows (b, t) of the global batch */
    c1v_seq dst;               /* apply_fwd: pooled / upsampled output; apply_bwd: gradient buffer of the conv output */
    uint8_t* sel_out;          /* apply_fwd POOL2: written */
} c1v_bn_args;
int c1v_bn_reduce(const c1v_bn_args*, int32_t backward, float* partial, int32_t blocks, float* sums, float sum_scale, void* stream);
int c1v_bn_stats(const float* sums, int32_t C, int64_t n_global, float eps, float momentum, float* mean, float* var,
                 float* moving_mean, float* moving_var, void* stream);
int c1v_bn_apply_fwd(const c1v_bn_args*, void* stream);
int c1v_bn_apply_bwd(const c1v_bn_args*, void* stream);

/* ---- a3+a4: latent head (:1188-1191, :1202, :1246, :1254) ---------------------------------------------------
 * mlv [B][2*D] f32 (z_mean | z_log_var, bias already added by the dense epilogue)
 * fwd: z = m + exp(std_scale*lv)*eps ; zd = keep_d ? z/keep_prob : 0 -> zd_out (T, [B][ldz]); z_out f32 optional
 *      kl_out[b] = mean_d(1 + lv - m^2 - exp(lv))
 * bwd: dz = dzd * keep_d/keep_prob; dm = dz + kl_w*2m/(D*Bglobal); dlv = dz*eps*std_scale*exp(std_scale*lv) + kl_w*(exp(lv)-1)/(D*Bglobal)
 *      -> dmlv (T, [B][ldd])  (std_scale = 1.0 is the reference) */
typedef struct c1v_latent_args {
    const float* mlv; int32_t B, D;
    const float* eps;            /* [B][D] */
    const uint8_t* keep_d;       /* [B][D] or NULL (inference) */
    float keep_prob; float std_scale;
    void* zd_out; int32_t dtype; int64_t ldz;
    float* z_out;                /* [B][D] or NULL */
    float* kl_out;               /* [B] or NULL */
    /* backward only */
    const float* dzd;            /* [B][D] f32 */
    int32_t B_global;
    void* dmlv; int64_t ldd;
    /* forward only, optional: mlv arrives as n_parts partial slabs (c1v_dense_fwd with k_splits) part_stride elements apart,
       starting at mlv; they are summed in slab order and the sum is written back to slab 0 (= mlv, then the complete
       [z_mean | z_log_var] the backward and transform() read) */
    int32_t n_parts; int64_t part_stride;
} c1v_latent_args;
int c1v_latent_head_fwd(const c1v_latent_args*, void* stream);
int c1v_latent_head_bwd(const c1v_latent_args*, void* stream);

/* latent head -> decoder dense -> first transposed convolution (:1246-1261) in ONE launch, for small batches: as three
 * launches they are 6.4 + 6.1 + 5.2 us at C2, nearly all of it launch and memory latency.  The arguments are those of the
 * three entry points (c1v_latent_head_fwd, c1v_dense_fwd, c1v_deconv1d_fwd), linked as the model links them: dense->a is
 * latent->zd_out, the dense epilogue (reshaping bf16 STORE) writes deconv->x, the deconv epilogue is a bf16 UP2 store.  Every
 * output of the three launches is written (the backward pass reads them), bit for bit the same.  bf16 only, shapes within
 * the kernel's limits (latent_dim <= 256, <= 8 partial slabs; dense K <= 256 and n_out <= 512; k * cp <= 128, n_out <= 128, n_rows <= 32):
 * c1v_dec_head_fwd_check returns 0 if a combination qualifies, without launching.  No counterpart in the reference. */
typedef struct c1v_dec_head_args {
    const c1v_latent_args* latent;
    const c1v_dense_args*  dense;
    const c1v_conv1d_args* deconv;
} c1v_dec_head_args;
int c1v_dec_head_fwd(const c1v_dec_head_args*, void* stream);
int c1v_dec_head_fwd_check(const c1v_dec_head_args*);

/* ---- a8: cosine-softmax cross-entropy against the vocabulary matrix (:1193-1203, :1319, :1326) --------------
 * Step 1  c1v_l2norm_rows: yhat = y / sqrt(max(sum y^2, 1e-12)) * qscale  -> T [M][kp] (zero padded), rnorm[M]
 * Step 2  c1v_cos_softmax_accum: for every row m and vocabulary row n:  e = exp2(yhat_m . w_n)   (qscale = log2(e)/tau)
 *           UT[s][m][c] = sum_{n in split s} e * w_n[c]   (c < kp; column `ones_col` of w holds 1.0 => UT[.][m][ones_col] = sum e)
 *         bf16: one fused MFMA kernel that never materialises the [M,N] logits; f32: NT GEMM + exp + TN GEMM
 *         through the caller's scratch `logits` ([n_chunk][m_pad] f32).
 * Step 3  c1v_cos_softmax_finish: per row: p_t, clip, xent, mask; per sample loss; dy = d loss / d y (T, padded conv-operand layout)
 */
typedef struct c1v_l2norm_args {
    const float* y; int32_t M, V; int64_t ldy;
    void* yhat; int32_t dtype; int32_t kp;
    float* rnorm;           /* 1/sqrt(max(sum y^2, eps)), indexed by the row of y */
    float qscale;
    const int32_t* pos;     /* optional (c1v_live_rows): row m of y lands in row pos[m] of yhat; pos[m] < 0: row skipped */
} c1v_l2norm_args;
int c1v_l2norm_rows(const c1v_l2norm_args*, void* stream);

typedef struct c1v_xent_accum_args {
    const void* yhat; int32_t dtype; int32_t M, kp;
    const void* w;    /* [n_pad][kp] T, rows >= N zero, column ones_col = 1.0 on rows < N */
    int32_t N, n_pad;
    void* ut;         /* [splits][m_pad][kp] f32 -- or bf16 when ut_dtype == C1V_BF16 (below) */
    int32_t m_pad, splits;
    int32_t need_u;   /* 0: only the sum-exp row `ones_col` is produced (validation loss) */
    int32_t ones_col;
    float* logits;    /* f32 path scratch [n_chunk][m_pad] */
    int32_t n_chunk;
    const int32_t* live_ctl;   /* optional DEVICE int[8] from c1v_live_rows (fused bf16 kernel only): yhat holds ctl[0] compacted
                                  rows; the grid of (m_pad/128) x splits workgroups is re-dealt on the device to ctl[1] row tiles
                                  x ctl[2] vocabulary splits, so ut must hold max_splits slabs (see c1v_live_rows) */
    int32_t ut_dtype;          /* C1V_F32 (default) or C1V_BF16: element type of the `ut` slabs.  bf16 slabs halve the bytes the
                                  finish step reads (splits x n_live x kp); only the fused bf16 kernel with need_u = 1 writes them */
} c1v_xent_accum_args;
int c1v_cos_softmax_accum(const c1v_xent_accum_args*, void* stream);

typedef struct c1v_xent_finish_args {
    const void* ut; int32_t splits, kp, m_pad, ones_col;
    const void* yhat; const void* w; int32_t dtype;
    const float* rnorm; float qscale;
    const int32_t* ids;       /* [B*T] target id or -1 */
    int32_t B, T, V;
    int32_t B_global;
    float* xent_rows;         /* [B*T] masked per-row cross entropy */
    float* xent_b;            /* [B] per-sample sum_t xent*mask / sum_t mask */
    c1v_seq dy;               /* or dy.data == NULL for loss only */
    const float* y; int64_t ldy;   /* optional fp32 decoder output [M][ldy]: yhat = y*rnorm in fp32 for the normalisation backward */
    const int32_t* pos;            /* optional: compact row of yhat / ut for sample row m (c1v_live_rows); < 0: no target */
    const int32_t* live_ctl;       /* optional: number of ut slabs = live_ctl[2] instead of `splits` */
    int32_t ut_dtype;              /* element type of the `ut` slabs, as given to c1v_cos_softmax_accum */
    const int32_t* live_rows; const int32_t* live_ids;   /* optional (c1v_live_rows_args::rows / cids, with pos and live_ctl): one
                                      wave per LIVE row -- compact row p < live_ctl[0] is sample row live_rows[p] with target
                                      live_ids[p]; rows without a target are not visited at all: their xent_rows are left alone
                                      (c1v_loss_reduce masks by ids) and their dy rows must be zero already (C1V_ROW_NORM's
                                      nrm_dy writes them) */
} c1v_xent_finish_args;
int c1v_cos_softmax_finish(const c1v_xent_finish_args*, void* stream);
/* Rows without a target (ids[m] < 0: padding behind the end of a text) contribute exactly zero to the loss and to its
 * gradient (reference :1199-1201 multiplies by the mask), so the loss stages may run on the compacted live rows only:
 *   pos[m] = index of row m among the rows with ids[m] >= 0, or -1
 *   ctl (DEVICE int[8]):
 *   ctl[0] = n_live, ctl[1] = ceil(n_live/128) row tiles,
 *   ctl[4] = FULL row tiles = the largest multiple of 256 (one workgroup per CU) <= ctl[1]: each takes the whole vocabulary
 *            in ONE workgroup (slab 0 of ut),
 *   ctl[2] = vocabulary splits of each of the remaining ctl[1] - ctl[4] row tiles, chosen so that they make up one more
 *            round of workgroups: clamp((ctl[4] ? 256 : items) / remaining, 1, min(max_splits, vocab_tiles)),
 *   ctl[3] = ceil(vocab_tiles / ctl[2])
 * `items` = workgroups of the c1v_cos_softmax_accum grid = (m_pad/128) * splits >= ctl[4] + remaining * ctl[2];
 * vocab_tiles = n_pad/64. */
int c1v_live_rows(const int32_t* ids, int32_t M, int32_t items, int32_t vocab_tiles, int32_t max_splits, int32_t* pos, int32_t* ctl, void* stream);
typedef struct c1v_live_rows_args {
    const int32_t* ids; int32_t M, items, vocab_tiles, max_splits; int32_t* pos; int32_t* ctl;
    int32_t* rows; int32_t* cids;   /* optional, both or neither: the inverse table rows[p] = m, cids[p] = ids[m] of the live row
                                       at position p (for c1v_xent_finish_args::live_rows) */
} c1v_live_rows_args;
int c1v_live_rows_table(const c1v_live_rows_args*, void* stream);     /* c1v_live_rows behind its argument struct (+ rows / cids) */
/* c1v_live_rows + c1v_gather_input as ONE launch (no counterpart in the reference: two independent jobs of a large batch's
 * step; block 0 builds the live-row table with 16 waves, the other blocks assemble the encoder operand) */
int c1v_live_rows_gather(const c1v_live_rows_args* live, const c1v_gather_input_args* gather, void* stream);
int c1v_cos_softmax_xent_fwd(const c1v_xent_accum_args*, void* stream);   /* need_u = 0 alias */
int c1v_cos_softmax_xent_bwd(const c1v_xent_accum_args*, void* stream);   /* need_u = 1 alias */

/* loss scalars (single block, fixed summation order):
 *   xent_b[b] = sum_t xent_rows[b,t] / #(ids[b,t] >= 0)           (:1200-1201)
 *   out3[0] = sum_b (xent_b[b] - kl_b[b]) / B_global  (:1203 + Keras batch mean); out3[1], out3[2] = the two terms */
int c1v_loss_reduce(const float* xent_rows, const int32_t* ids, const float* kl_b, int32_t B, int32_t T, int32_t B_global,
                    float* out3, float* xent_b, void* stream);

/* ---- a9: Nadam(clipnorm) over the flat parameter buffer (:1329) ----------------------------------------------
 * c1v_grad_finalize: grad[seg.offset + i] = sum_{s<splits} sum_{r<reps} slab[s*slab_stride + r*rep_stride + i]
 *                    and sq_partial[block] = partial sum of grad^2 (when sq_partial != NULL)
 * c1v_sqnorm_partial: partial sums of squares of an arbitrary f32 buffer (after an all-reduce)
 * c1v_nadam_step:    Keras-2.2 Nadam; clip scale = clipnorm/norm if norm >= clipnorm, norm = sqrt(sched[7])*grad_scale */
typedef struct c1v_seg_desc { const float* slab; int64_t slab_stride; int32_t splits; int32_t reps; int64_t rep_stride; int64_t offset; int64_t count; int32_t frozen; int32_t first_block; } c1v_seg_desc;
/* segment i owns blocks [first_block_i, first_block_{i+1}) (ceil(count / C1V_CHUNK) each); sq_partial has total_blocks entries */
int c1v_grad_finalize(const c1v_seg_desc* segs /*device*/, int32_t n_seg, int32_t total_blocks, float* grad, float* sq_partial, void* stream);
/* c1v_grad_finalize plus the step's loss scalars (c1v_loss_reduce, same arithmetic and order) in one extra workgroup of
 * the same launch: both close the backward pass and the scalars are off the critical path. */
typedef struct c1v_loss_reduce_args {
    const float* xent_rows; const int32_t* ids; const float* kl_b; int32_t B, T, B_global; float* out3; float* xent_b;
    int32_t pre_summed;   /* 1: xent_b already holds the per-sample means (c1v_loss_sample_means); only the sums over b are left */
} c1v_loss_reduce_args;
/* the per-sample part of c1v_loss_reduce as a launch of its own, one wave per sample: xent_b[b] = sum_t xent_rows[b,t] / #targets.
 * For large batches (C4: 270 k rows), where the single workgroup of c1v_grad_finalize_loss was twice as long as the rest of
 * that launch. */
int c1v_loss_sample_means(const c1v_loss_reduce_args*, void* stream);
int c1v_grad_finalize_loss(const c1v_seg_desc* segs /*device*/, int32_t n_seg, int32_t total_blocks, float* grad, float* sq_partial,
                           const c1v_loss_reduce_args* loss, void* stream);
/* both of the above behind ONE argument struct (loss == NULL: no loss scalars), so that the launch can be an entry of
 * c1v_run_list */
typedef struct c1v_grad_finalize_args {
    const c1v_seg_desc* segs; int32_t n_seg, total_blocks; float* grad; float* sq_partial; const c1v_loss_reduce_args* loss;
} c1v_grad_finalize_args;
int c1v_grad_finalize_step(const c1v_grad_finalize_args*, void* stream);
int c1v_sqnorm_partial(const float* x, int64_t n, float* sq_partial, int32_t n_partial, void* stream);
typedef struct c1v_nadam_args {
    float* p; const float* g; float* m; float* v; int64_t n;
    float grad_scale;          /* 1/world_size after a sum all-reduce */
    float clipnorm;
    const float* sched;        /* DEVICE [8]: lr, mu_t, mu_t1, 1-m_schedule_new, 1-m_schedule_next, 1-beta2^t, t, sum of squares of the gradient */
    float beta1, beta2, eps;
    float* norm_out;           /* optional [1] */
    const float* sq_partial;   /* optional: sum of squares = sum of sq_partial[0..n_partial) computed inside this kernel instead */
    int32_t n_partial;         /*           of sched[7] (sched then only needs the schedule scalars, see c1v_step_prologue) */
    float* sched_out7;         /*           required with sq_partial: receives that sum (normally &sched[7]) */
    uint64_t* ctr;             /* optional: RNG step counter, incremented once (c1v_nadam_advance does it otherwise) */
} c1v_nadam_args;
int c1v_nadam_step(const c1v_nadam_args*, void* stream);
/* state (DEVICE double[3]) = {iterations, m_schedule, lr}; writes the scalars of the NEXT step into sched[8] (sched[7] =
 * sum of sq_partial[0..n_partial)) and advances the state and, if given, the RNG step counter, so that a captured HIP
 * graph can be replayed without host-side scalar updates. */
int c1v_nadam_advance(double* state, float* sched, const float* sq_partial, int32_t n_partial, uint64_t* ctr, void* stream);

/* ---- a10-i: EOS truncation + renormalisation (:542-565) ------------------------------------------------------
 * y [B][T][V] f32 -> out [B][T][V-1] f32, end_idx[B] */
int c1v_eos_truncate_normalize(const float* y, int32_t B, int32_t T, int32_t V, float* out, int32_t* end_idx, void* stream);
/* the same on rows that lie `ldy` >= V values apart (the engine pads the decoder output's rows to whole 16-byte words) */
int c1v_eos_truncate_normalize_ld(const float* y, int64_t ldy, int32_t B, int32_t T, int32_t V, float* out, int32_t* end_idx, void* stream);
/* float64 variant: the reference's static helper keeps the dtype of its NumPy argument (:547-550) and its known-answer
 * tests compare float64 values to 7 places (tests/test_conv1d_text_vae.py:1544-1605) */
int c1v_eos_truncate_normalize_f64(const double* y, int32_t B, int32_t T, int32_t V, double* out, int32_t* end_idx, void* stream);

/* ---- a10-ii: nearest vocabulary row (gensim similar_by_vector arithmetic, :587) -------------------------------
 * score[m] = max_n q_m . w_n, idx[m] = lowest n attaining it (n_base added), over the caller's shard of rows.
 * bf16: fused MFMA GEMM + arg-max; f32: NT GEMM into `logits` + arg-max kernel. */
typedef struct c1v_vocab_argmax_args {
    const void* q; int32_t dtype; int32_t M, kp;
    const void* w; int32_t N, n_pad; int32_t n_base;
    float* part_score; int32_t* part_idx;   /* [splits][m_pad] scratch */
    int32_t m_pad, splits;
    float* score; int32_t* idx;             /* [M] */
    float* logits; int32_t n_chunk;
} c1v_vocab_argmax_args;
int c1v_vocab_argmax(const c1v_vocab_argmax_args*, void* stream);
/* topn > 1 (`similar_by_vector(v, topn)` inside find_best_words, :587): NT GEMM into the caller's `logits` scratch
 * ([M][ld] f32, ld >= N) + per-row selection; score/idx are [M][topn], best first, ties -> lowest index.  q [M][kp] and
 * w [>= N rows][kp] follow the operand rules of c1v_dense_fwd (rows zero-padded to whole 128-byte K tiles, 64 readable
 * rows past the end). */
typedef struct c1v_vocab_topn_args {
    const void* q; int32_t dtype; int32_t M, kp;
    const void* w; int32_t N; int32_t n_base;
    float* logits; int64_t ld;
    int32_t topn;
    float* score; int32_t* idx;
} c1v_vocab_topn_args;
int c1v_vocab_topn(const c1v_vocab_topn_args*, void* stream);
/* c1v_vocab_argmax (bf16, fused kernel) for callers that CANNOT ALLOCATE device memory -- code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference (which has no device
 * code at all).  The caller passes only inputs and outputs; the per-split partial (score, index) rows, which
 * c1v_vocab_argmax takes as caller scratch, live in a static device array of C1V_NOALLOC_SCRATCH_BYTES inside the library
 * (one copy per GPU, held until the process ends).  Limits: bf16 only; kp in {32, 64, 320}; at most
 * C1V_NOALLOC_MAX_QUERIES rows per call (more is rejected with an error, nothing is launched); q [M][kp] bf16 and
 * w [n_pad][kp] bf16 with rows >= N zero, as for c1v_vocab_argmax.  Results are identical to c1v_vocab_argmax's.
 * The call allocates nothing and does not synchronise; the FIRST call on a device looks the array's address up
 * (hipGetSymbolAddress), so make that call outside a graph capture.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (two streams, two threads) race -- serialise them. */
#define C1V_NOALLOC_SCRATCH_BYTES (8u << 20)
#define C1V_NOALLOC_MAX_QUERIES   ((int32_t)(C1V_NOALLOC_SCRATCH_BYTES / 8 / 128 * 128))
#define C1V_NOALLOC_MAX_DEVICES   64
typedef struct c1v_vocab_argmax_noalloc_args {
    const void* q; int32_t M, kp;
    const void* w; int32_t N, n_pad; int32_t n_base;
    float* score; int32_t* idx;             /* [M] */
} c1v_vocab_argmax_noalloc_args;
int c1v_vocab_argmax_noalloc(const c1v_vocab_argmax_noalloc_args*, void* stream);

/* Up to C1V_WGRAD_BATCH_MAX weight gradients (conv / deconv / dense, tap_group 0, one operand type) in ONE launch.  For small
 * batches (C2): every wgrad of the model is a few dozen workgroups and as long as its longest one, none of them feeds the
 * chain of data gradients, so the backward pass becomes that chain (dgrad launches alone are 7-9 us where the paired
 * dgrad + wgrad launches were 9-15) plus one launch for all weight gradients.  Results are those of the single launches,
 * bit for bit.  No counterpart in the reference. */
#define C1V_WGRAD_BATCH_MAX 12
typedef struct c1v_wgrad_batch_args { int32_t n; const c1v_wgrad_args* items[C1V_WGRAD_BATCH_MAX]; } c1v_wgrad_batch_args;
int c1v_wgrad_batch(const c1v_wgrad_batch_args*, void* stream);

/* A recorded step: calls fns[i](args[i], stream) for i = 0 .. n-1 and stops at the first non-zero return code (which it
 * returns).  Every entry point above of the form `int f(const X_args*, void* stream)` can be listed.  Why: a training step at
 * C2 is 16 launches of 5-50 us; issued one by one from Python the host is the bottleneck, replayed as a HIP graph the runtime
 * leaves ~8 us between two replays (4 % of the step, profiles/r03c_c2_trace_step_graph.txt), issued back to back from native code
 * there is no gap.  No counterpart in the reference (Keras' session.run plays this role there). */
typedef int (*c1v_op_fn)(const void* args, void* stream);
int c1v_run_list(int32_t n, const c1v_op_fn* fns, const void* const* args, void* stream);
/* ... with a second stream for entries that are off the critical path (the weight gradients of a small-batch step run
 * next to the chain of data gradients, which leaves most CUs idle).  flags[i]: C1V_LIST_SIDE = issue entry i on `side`;
 * C1V_LIST_FORK = (side entry) it first waits for everything issued on `stream` so far; C1V_LIST_JOIN = (main entry) it
 * first waits for everything issued on `side`.  Unjoined side work is joined after the last entry: when the call returns,
 * everything it issued is ordered on `stream`.  Not for use under stream capture.  Measured at C2: every cross-stream edge
 * costs ~9 us, more than the overlap hides, so the engine does not use it by default (C1V_WGRAD_OVERLAP). */
#define C1V_LIST_SIDE 1
#define C1V_LIST_FORK 2
#define C1V_LIST_JOIN 4
int c1v_run_list_fork(int32_t n, const c1v_op_fn* fns, const void* const* args, const int32_t* flags, void* stream, void* side);
/* merge per-shard results gathered from R ranks: ties -> lowest global index */
int c1v_argmax_merge(const float* scores /*[R][M]*/, const int32_t* idxs, int32_t R, int32_t M, float* score, int32_t* idx, void* stream);

/* ---- counter-based RNG (Philox4x32-10): element e of stream uses block e>>2, word e&3 ------------------------
 * out[i] is element (offset + ctr[0]*ctr_mul + i) of stream `stream_id` under key `seed` (ctr may be NULL).  The
 * device-side counter lets a captured HIP graph draw fresh numbers on every replay; keying by the GLOBAL element
 * index makes 1-GPU and N-GPU runs draw the same numbers for the same global sample. */
int c1v_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t offset,
                      const uint64_t* ctr, uint64_t ctr_mul, void* stream);
int c1v_philox_bernoulli(uint8_t* out, int64_t n, float keep_prob, uint64_t seed, uint64_t stream_id, uint64_t offset,
                         const uint64_t* ctr, uint64_t ctr_mul, void* stream);
int c1v_counter_add(uint64_t* ctr, uint64_t inc, void* stream);
/* several requests in ONE launch: kind 0 = normal (float out), 1 = bernoulli (uint8 out, thr = keep_prob * 2^24);
 * request i owns blocks [first_block_i, first_block_{i+1}) with ceil((n + 3) / 1024) + 1 blocks each */
typedef struct c1v_rng_desc { void* out; int64_t n; int32_t kind; uint32_t thr; uint64_t stream_id; uint64_t offset; uint64_t ctr_mul; int32_t first_block; int32_t pad; } c1v_rng_desc;
int c1v_philox_multi(const c1v_rng_desc* descs /*device*/, int32_t n_desc, int32_t total_blocks, uint64_t seed, const uint64_t* ctr, void* stream);

/* Step prologue: the jobs of a step that depend on nothing but its inputs, in ONE launch -- c1v_philox_multi (n_rng > 0),
 * c1v_pack_input or c1v_gather_input (at most one of pack / gather) and c1v_live_rows (live != NULL); every part is
 * optional and computes exactly what the separate entry point computes.  No counterpart in the reference: Keras draws
 * its dropout masks and builds batches op by op; here each separate launch cost 5-11 us of a ~260 us captured step. */
typedef struct c1v_step_prologue_args {
    const c1v_rng_desc* rng_descs; int32_t n_rng; int32_t rng_blocks; uint64_t seed; const uint64_t* ctr;
    const c1v_pack_input_args* pack;
    const c1v_gather_input_args* gather;
    const c1v_live_rows_args* live;
    double* nadam_state; float* sched;   /* optional: the schedule scalars of this step's Nadam update (c1v_nadam_advance
                                            without the gradient norm and without the RNG-counter increment; pair it with
                                            c1v_nadam_step's sq_partial / ctr fields) */
    const c1v_pack_desc* pack_descs; int32_t n_pack; int32_t pack_blocks;   /* optional: c1v_pack_weights (the operand copies
                                            of the weights the PREVIOUS step updated: refreshed here, beside the other jobs,
                                            instead of in a launch of their own at the end of that step) */
} c1v_step_prologue_args;
int c1v_step_prologue(const c1v_step_prologue_args*, void* stream);

/* ---- misc ---------------------------------------------------------------------------------------------------- */
const char* c1v_last_error(void);
int c1v_version(void);
/* info[0]=CU count, [1]=LDS bytes/CU, [2]=wavefront size, [3]=gfx arch number (950) */
int c1v_device_info(int32_t device, int32_t* info4);

#ifdef __cplusplus
}
#endif
#endif
