// BatchNormalization of the optional `use_batch_norm=True` graph (reference :1229-1230, :1238-1239, :1259-1260,
// :1270-1272; Keras 2.2 defaults: axis -1, momentum 0.99, epsilon 1e-3) as HBM-bound kernels around the conv GEMMs.
//
// The layer sits between a convolution's ReLU and the MaxPool1D(2) (encoder) or UpSampling1D(2) (decoder) behind it, and
// its training-phase statistics couple every (sample, step) of the batch, so the fused conv epilogues cannot be used:
//   forward   conv GEMM (bias + ReLU, plain store) -> bn_reduce (sum x, sum x^2 per channel, two deterministic stages)
//             -> bn_stats (mean, variance, moving averages) -> bn_apply_fwd (normalise + pool / upsample / dropout)
//   backward  next layer's dgrad (plain store) -> bn_reduce (sum dy, sum dy*xhat; the pool / upsample backward is applied
//             on the fly) -> bn_apply_bwd (normalisation backward + the ReLU mask) -> this layer's dgrad / wgrad GEMMs
// Sums are plain sums so that data-parallel ranks can all-reduce them (statistics over the GLOBAL batch: results do not
// depend on the GPU count).  No float atomics: partial sums per block, then a fixed-order second stage.
#include "c1v_common.h"

constexpr int BN_TC = 64;            // lanes over channels
constexpr int BN_TR = 4;             // row lanes
constexpr int BN_CPT = 8;            // channels per thread per pass (64 * 8 = 512 channels per pass)

struct BnView {                      // device copy of c1v_seq + derived
    const void* data; int dtype; int64_t bstride; int off, len, cp, c;
};
static BnView bn_view(const c1v_seq& s) { return BnView{s.data, s.dtype, s.bstride, s.off, s.len, s.cp, s.c}; }
__device__ __forceinline__ float bn_ld(const BnView& v, int b, int t, int c) {
    return load_any(v.data, v.dtype, (int64_t)b * v.bstride + (int64_t)(v.off + t) * v.cp + c);
}

struct BnDev {
    BnView x;                        // post-ReLU conv output [B][L][C] (the normalisation's input)
    int B, L, C;
    int mode;                        // C1V_BN_POOL2 / C1V_BN_UP2
    // gradient source (backward): POOL2: dP [B][L/2][C] (+ sel, + optional keep mask of the encoder dropout);
    //                             UP2:   dU [B][up_len][C] (gradient of the upsampled, cropped output)
    BnView g; const uint8_t* sel; int64_t sel_bstride; int sel_cp; const uint8_t* keep; float keep_scale;
    const float* mean; const float* var; float eps;       // statistics the normalisation uses
    const float* gamma; const float* beta;
    const float* sums;               // backward: [2][C] = sum dy, sum dy*xhat (already over the global batch)
    float inv_n;                     // 1 / (global rows)
};

// gradient w.r.t. the normalisation's OUTPUT at (b, t, c): MaxPool1D / UpSampling1D backward on the fly
__device__ __forceinline__ float bn_dy_at(const BnDev& a, int b, int t, int c) {
    if (a.mode == C1V_BN_POOL2) {
        const int t2 = t >> 1;
        if (t2 >= a.g.len) return 0.0f;                                  // odd tail row dropped by the pool
        const int64_t si = (int64_t)b * a.sel_bstride + (int64_t)t2 * a.sel_cp + c;
        if ((a.sel[si] != 0) != ((t & 1) != 0)) return 0.0f;
        float g = bn_ld(a.g, b, t2, c);
        if (a.keep) g = a.keep[si] ? g * a.keep_scale : 0.0f;
        return g;
    }
    float g = 0.0f;
    if (2 * t < a.g.len) g += bn_ld(a.g, b, 2 * t, c);
    if (2 * t + 1 < a.g.len) g += bn_ld(a.g, b, 2 * t + 1, c);
    return g;
}

// ---- stage 1: per-block partial sums.  BWD = false: (x, x^2); BWD = true: (dy, dy * xhat) ----------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_kernel(BnDev a, float* __restrict__ partial /*[blocks][2][C]*/) {
    __shared__ float red[2][BN_TR][BN_TC];
    const int tx = threadIdx.x % BN_TC, ty = threadIdx.x / BN_TC;
    const int64_t rows = (int64_t)a.B * a.L;
    const int64_t per = c1v_cdiv(rows, (int64_t)gridDim.x);
    const int64_t r0 = (int64_t)blockIdx.x * per;
    int64_t r1 = r0 + per; if (r1 > rows) r1 = rows;
    for (int c0 = 0; c0 < a.C; c0 += BN_TC * BN_CPT) {
        float s1[BN_CPT], s2[BN_CPT], mu[BN_CPT], rs[BN_CPT];
#pragma unroll
        for (int k = 0; k < BN_CPT; ++k) {
            s1[k] = 0.f; s2[k] = 0.f;
            const int c = c0 + tx + BN_TC * k;
            mu[k] = 0.f; rs[k] = 0.f;
            if (BWD && c < a.C) { mu[k] = a.mean[c]; rs[k] = rsqrtf(a.var[c] + a.eps); }
        }
        for (int64_t r = r0 + ty; r < r1; r += BN_TR) {
            const int b = (int)(r / a.L), t = (int)(r - (int64_t)b * a.L);
#pragma unroll
            for (int k = 0; k < BN_CPT; ++k) {
                const int c = c0 + tx + BN_TC * k;
                if (c < a.C) {
                    const float x = bn_ld(a.x, b, t, c);
                    if (BWD) { const float dy = bn_dy_at(a, b, t, c); s1[k] += dy; s2[k] += dy * ((x - mu[k]) * rs[k]); }
                    else { s1[k] += x; s2[k] += x * x; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < BN_CPT; ++k) {
            __syncthreads();
            red[0][ty][tx] = s1[k]; red[1][ty][tx] = s2[k];
            __syncthreads();
            const int c = c0 + tx + BN_TC * k;
            if (ty == 0 && c < a.C) {
                partial[((int64_t)blockIdx.x * 2 + 0) * a.C + c] = (red[0][0][tx] + red[0][1][tx]) + (red[0][2][tx] + red[0][3][tx]);
                partial[((int64_t)blockIdx.x * 2 + 1) * a.C + c] = (red[1][0][tx] + red[1][1][tx]) + (red[1][2][tx] + red[1][3][tx]);
            }
        }
    }
}

// ---- stage 2: partials -> sums [2][C], in block order ------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_sum_partials_kernel(const float* __restrict__ partial, int blocks, int C, float* __restrict__ sums, float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // index into [2][C]
    if (i >= 2 * C) return;
    const int which = i / C, c = i - which * C;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += partial[((int64_t)b * 2 + which) * C + c];
    sums[i] = s * scale;
}

// ---- statistics from the (global) sums; moving averages (Keras: momentum 0.99, variance * n / (n - (1 + eps))) ------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ sums, int C, float n_global, float eps, float momentum,
                                                       float* __restrict__ mean, float* __restrict__ var,
                                                       float* __restrict__ moving_mean, float* __restrict__ moving_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float m = sums[c] / n_global;
    float v = sums[C + c] / n_global - m * m;
    v = v > 0.f ? v : 0.f;
    mean[c] = m; var[c] = v;
    if (moving_mean) {
        moving_mean[c] = moving_mean[c] * momentum + m * (1.0f - momentum);
        moving_var[c] = moving_var[c] * momentum + v * (n_global / (n_global - (1.0f + eps))) * (1.0f - momentum);
    }
}

// ---- forward: y = gamma * (x - mean) * rsqrt(var + eps) + beta, then MaxPool1D(2) (+ sel, + dropout) or UpSampling1D(2) --
struct BnFwdDst { void* data; int dtype; int64_t bstride; int off, len, cp; uint8_t* sel; int64_t sel_bstride; int sel_cp; const uint8_t* keep; float keep_scale; };

__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(BnDev a, BnFwdDst d) {
    const int rows_out = a.mode == C1V_BN_POOL2 ? a.L / 2 : a.L;
    const int64_t total = (int64_t)a.B * rows_out * a.C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.C);
        const int64_t r = i / a.C;
        const int b = (int)(r / rows_out), t = (int)(r - (int64_t)b * rows_out);
        const float sc = a.gamma[c] * rsqrtf(a.var[c] + a.eps), sh = a.beta[c] - a.mean[c] * sc;
        if (a.mode == C1V_BN_POOL2) {
            if (t >= d.len) continue;
            const float y0 = bn_ld(a.x, b, 2 * t, c) * sc + sh, y1 = bn_ld(a.x, b, 2 * t + 1, c) * sc + sh;
            const bool s1 = y1 > y0;                                 // first maximum wins ties
            float y = s1 ? y1 : y0;
            const int64_t si = (int64_t)b * d.sel_bstride + (int64_t)t * d.sel_cp + c;
            if (d.sel) d.sel[si] = s1 ? 1 : 0;
            if (d.keep) y = d.keep[si] ? y * d.keep_scale : 0.0f;
            store_any(d.data, d.dtype, (int64_t)b * d.bstride + (int64_t)(d.off + t) * d.cp + c, y);
        } else {
            const float y = bn_ld(a.x, b, t, c) * sc + sh;
            const int64_t o = (int64_t)b * d.bstride + (int64_t)(d.off + 2 * t) * d.cp + c;
            if (2 * t < d.len) store_any(d.data, d.dtype, o, y);
            if (2 * t + 1 < d.len) store_any(d.data, d.dtype, o + d.cp, y);
        }
    }
}

// ---- backward: dx = gamma * rstd * (dy - sum_dy / n - xhat * sum_dy_xhat / n), times the ReLU mask of the conv in front ----
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(BnDev a, BnFwdDst d) {
    const int64_t total = (int64_t)a.B * a.L * a.C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.C);
        const int64_t r = i / a.C;
        const int b = (int)(r / a.L), t = (int)(r - (int64_t)b * a.L);
        const float x = bn_ld(a.x, b, t, c);
        const float rs = rsqrtf(a.var[c] + a.eps);
        const float xhat = (x - a.mean[c]) * rs;
        const float dy = bn_dy_at(a, b, t, c);
        float dx = a.gamma[c] * rs * (dy - a.sums[c] * a.inv_n - xhat * (a.sums[a.C + c] * a.inv_n));
        if (!(x > 0.0f)) dx = 0.0f;                                   // ReLU of the convolution (stored activation is post-ReLU)
        store_any(d.data, d.dtype, (int64_t)b * d.bstride + (int64_t)(d.off + t) * d.cp + c, dx);
    }
}

// ------------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------------
static unsigned bn_grid(int64_t n) {
    int64_t g = c1v_cdiv(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

static int bn_fill(const c1v_bn_args* a, BnDev& d, const char* what) {
    C1V_REQUIRE(a && a->x.data, "%s: null argument", what);
    C1V_REQUIRE(a->x.batch > 0 && a->x.len > 0 && a->x.c > 0 && a->x.c <= a->x.cp, "%s: bad activation descriptor", what);
    C1V_REQUIRE(a->mode == C1V_BN_POOL2 || a->mode == C1V_BN_UP2, "%s: bad mode %d", what, a->mode);
    d = BnDev{};
    d.x = bn_view(a->x); d.B = a->x.batch; d.L = a->x.len; d.C = a->x.c; d.mode = a->mode;
    d.g = bn_view(a->grad); d.sel = a->sel; d.sel_bstride = a->sel_bstride; d.sel_cp = a->sel_cp; d.keep = a->keep; d.keep_scale = a->keep_scale;
    d.mean = a->mean; d.var = a->var; d.eps = a->eps; d.gamma = a->gamma; d.beta = a->beta; d.sums = a->sums;
    d.inv_n = a->n_global > 0 ? 1.0f / (float)a->n_global : 0.0f;
    return 0;
}

extern "C" int c1v_bn_reduce(const c1v_bn_args* a, int32_t backward, float* partial, int32_t blocks, float* sums, float sum_scale, void* stream) {
    BnDev d;
    if (bn_fill(a, d, "bn_reduce")) return -1;
    C1V_REQUIRE(partial && sums && blocks > 0 && blocks <= 4096, "bn_reduce: bad partial buffer (blocks=%d)", blocks);
    if (backward) {
        C1V_REQUIRE(a->grad.data && a->mean && a->var, "bn_reduce: backward needs the gradient source and the statistics");
        C1V_REQUIRE(a->mode != C1V_BN_POOL2 || a->sel, "bn_reduce: POOL2 backward needs sel");
        C1V_REQUIRE(a->grad.c >= d.C, "bn_reduce: gradient buffer has fewer channels than the activation");
        hipLaunchKernelGGL(bn_reduce_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d, partial);
    } else {
        hipLaunchKernelGGL(bn_reduce_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d, partial);
    }
    C1V_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_sum_partials_kernel, dim3((unsigned)c1v_cdiv(2 * d.C, 256)), dim3(256), 0, (hipStream_t)stream, partial, blocks, d.C, sums, sum_scale);
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_bn_stats(const float* sums, int32_t C, int64_t n_global, float eps, float momentum, float* mean, float* var,
                            float* moving_mean, float* moving_var, void* stream) {
    C1V_REQUIRE(sums && mean && var && C > 0 && n_global > 1, "bn_stats: bad argument");
    C1V_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "bn_stats: both moving averages or neither");
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)c1v_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, C, (float)n_global, eps, momentum,
                       mean, var, moving_mean, moving_var);
    C1V_LAUNCH_CHECK();
    return 0;
}

static int bn_dst(const c1v_bn_args* a, BnFwdDst& o, const char* what) {
    C1V_REQUIRE(a->dst.data && a->dst.len > 0 && a->dst.cp >= a->x.c && a->dst.batch == a->x.batch, "%s: bad destination", what);
    o = BnFwdDst{a->dst.data, a->dst.dtype, a->dst.bstride, a->dst.off, a->dst.len, a->dst.cp, a->sel_out, a->sel_bstride, a->sel_cp, a->keep, a->keep_scale};
    return 0;
}

extern "C" int c1v_bn_apply_fwd(const c1v_bn_args* a, void* stream) {
    BnDev d;
    BnFwdDst o;
    if (bn_fill(a, d, "bn_apply_fwd") || bn_dst(a, o, "bn_apply_fwd")) return -1;
    C1V_REQUIRE(a->mean && a->var && a->gamma && a->beta, "bn_apply_fwd: needs statistics, gamma and beta");
    C1V_REQUIRE(a->mode != C1V_BN_POOL2 || a->dst.len <= a->x.len / 2, "bn_apply_fwd: pooled destination longer than len / 2");
    const int rows_out = a->mode == C1V_BN_POOL2 ? d.L / 2 : d.L;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(bn_grid((int64_t)d.B * rows_out * d.C)), dim3(256), 0, (hipStream_t)stream, d, o);
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_bn_apply_bwd(const c1v_bn_args* a, void* stream) {
    BnDev d;
    BnFwdDst o;
    if (bn_fill(a, d, "bn_apply_bwd") || bn_dst(a, o, "bn_apply_bwd")) return -1;
    C1V_REQUIRE(a->grad.data && a->mean && a->var && a->gamma && a->sums && a->n_global > 0, "bn_apply_bwd: missing argument");
    C1V_REQUIRE(a->mode != C1V_BN_POOL2 || a->sel, "bn_apply_bwd: POOL2 needs sel");
    C1V_REQUIRE(a->dst.len >= a->x.len, "bn_apply_bwd: destination shorter than the activation");
    o.sel = nullptr; o.keep = nullptr;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(bn_grid((int64_t)d.B * d.L * d.C)), dim3(256), 0, (hipStream_t)stream, d, o);
    C1V_LAUNCH_CHECK();
    return 0;
}
