// Loss scalars of a step (shared by c1v_loss_reduce and the extra block of c1v_grad_finalize_loss).
#pragma once
#include "c1v_common.h"

// loss scalars (single block, deterministic order)
constexpr int LR_CAP = 8448;                       // row values staged per pass (256 samples of 32 steps at pitch 33)
// (one workgroup of any whole number of waves <= 16; also run as an extra block of grad_finalize)
__device__ __forceinline__ void loss_reduce_body(const float* __restrict__ xent_rows, const int* __restrict__ ids,
                                                 const float* __restrict__ kl_b, int B, int T_len, int B_global,
                                                 float* __restrict__ out3, float* __restrict__ xent_b) {
    // Pass structure: all threads stage the per-step values of S = LR_CAP / (T + 1) samples in LDS with independent,
    // coalesced loads (a wave-per-sample loop serialised one global-load latency per sample: 16 us at B = 256), then
    // thread i sums the T steps of sample i in step order (pitch T + 1: conflict-free).  Every sum has a fixed order.
    __shared__ float vx[LR_CAP];
    __shared__ uint8_t vc[LR_CAP];
    __shared__ float sx[16], sk[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
    float ax = 0.f, ak = 0.f;
    if (T_len <= LR_CAP / 2) {
        const int pitch = T_len + 1, S = LR_CAP / pitch;
        const float rcp = 1.0f / (float)T_len;
        for (int base = 0; base < B; base += S) {
            const int ns = B - base < S ? B - base : S, n = ns * T_len;
            const float* xr = xent_rows + (int64_t)base * T_len;
            const int* ir = ids + (int64_t)base * T_len;
            const bool vec = (n & 3) == 0 && (((((uintptr_t)xr) | ((uintptr_t)ir)) & 15) == 0);
            if (vec) {
                // 16-byte words: four times fewer (independent) requests per thread
#pragma unroll 8
                for (int i4 = tid; i4 < (n >> 2); i4 += nthr) {
                    const float4 xv = *(const float4*)(xr + 4 * i4);
                    const int4 iv = *(const int4*)(ir + 4 * i4);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
                    const int is[4] = {iv.x, iv.y, iv.z, iv.w};
                    int bl = (int)(((float)(4 * i4) + 0.5f) * rcp);
                    int t = 4 * i4 - bl * T_len;
                    if (t < 0) { --bl; t += T_len; } else if (t >= T_len) { ++bl; t -= T_len; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        vx[bl * pitch + t] = xs[k];
                        vc[bl * pitch + t] = is[k] >= 0 ? 1 : 0;
                        if (++t == T_len) { t = 0; ++bl; }
                    }
                }
            } else {
#pragma unroll 4
                for (int i = tid; i < n; i += nthr) {
                    int bl = (int)(((float)i + 0.5f) * rcp);
                    int t = i - bl * T_len;
                    if (t < 0) { --bl; t += T_len; } else if (t >= T_len) { ++bl; t -= T_len; }
                    vx[bl * pitch + t] = xr[i];
                    vc[bl * pitch + t] = ir[i] >= 0 ? 1 : 0;
                }
            }
            __syncthreads();
            for (int bl = tid; bl < ns; bl += nthr) {
                float s = 0.f, c = 0.f;
                for (int t = 0; t < T_len; ++t) { s += vx[bl * pitch + t]; c += (float)vc[bl * pitch + t]; }
                const float xb = s / c;
                if (xent_b) xent_b[base + bl] = xb;
                ax += xb;
                ak += kl_b ? kl_b[base + bl] : 0.f;
            }
            __syncthreads();
        }
        ax = wave_sum(ax); ak = wave_sum(ak);
    } else {
        // very long sequences: one wave per sample, lanes over the steps
        for (int b = wave; b < B; b += nwave) {
            float s = 0.f, c = 0.f;
            for (int t = lane; t < T_len; t += 64) { s += xent_rows[(int64_t)b * T_len + t]; c += ids[(int64_t)b * T_len + t] >= 0 ? 1.f : 0.f; }
            s = wave_sum(s); c = wave_sum(c);
            const float xb = s / c;
            if (lane == 0 && xent_b) xent_b[b] = xb;
            ax += xb;
            ak += kl_b ? kl_b[b] : 0.f;
        }
    }
    if (lane == 0) { sx[wave] = ax; sk[wave] = ak; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tx = 0.f, tk = 0.f;
        for (int i = 0; i < nwave; ++i) { tx += sx[i]; tk += sk[i]; }
        const float inv = 1.0f / (float)B_global;
        out3[0] = (tx - tk) * inv; out3[1] = tx * inv; out3[2] = tk * inv;
    }
}


constexpr int VS_BM_ROWS = 128;   // sample rows per workgroup of the vocabulary stream kernel

// positions of the rows that have a target (ids[m] >= 0) in compact order, -1 for the others, and the work split of the
// vocabulary stream over them:
//   ctl[0] = n_live, ctl[1] = live row tiles (128 rows), ctl[2] = vocabulary splits per row tile, ctl[3] = tiles per split
// The stream kernels are launched with a fixed grid of `items` workgroups (a captured graph cannot resize it); with fewer
// live row tiles each one is cut into more vocabulary splits so that the grid still covers the chip.
// One workgroup of any whole number of waves <= 16 (also run as a block of c1v_step_prologue): every thread owns a run of
// consecutive rows (a multiple of 4, read and written as 16-byte words), the runs are scanned through LDS.
__device__ __forceinline__ void live_rows_body(const int* __restrict__ ids, int M, int items, int vocab_tiles, int max_splits,
                                               int* __restrict__ pos, int* __restrict__ ctl) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
    const int per = ((M + nthr - 1) / nthr + 3) & ~3;
    const int m0 = tid * per;
    const bool vec = ((((uintptr_t)ids) | ((uintptr_t)pos)) & 15) == 0;
    int cnt = 0;
    for (int i = 0; i < per; i += 4) {
        const int m = m0 + i;
        if (vec && m + 3 < M) {
            const int4 v = *(const int4*)(ids + m);
            cnt += (v.x >= 0) + (v.y >= 0) + (v.z >= 0) + (v.w >= 0);
        } else {
            for (int k = 0; k < 4; ++k) cnt += (m + k < M && ids[m + k] >= 0) ? 1 : 0;
        }
    }
    int incl = cnt;                                    // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int p = base + incl - cnt;
    for (int i = 0; i < per; i += 4) {
        const int m = m0 + i;
        if (vec && m + 3 < M) {
            const int4 v = *(const int4*)(ids + m);
            int4 o;
            o.x = v.x >= 0 ? p++ : -1; o.y = v.y >= 0 ? p++ : -1; o.z = v.z >= 0 ? p++ : -1; o.w = v.w >= 0 ? p++ : -1;
            *(int4*)(pos + m) = o;
        } else {
            for (int k = 0; k < 4; ++k)
                if (m + k < M) pos[m + k] = ids[m + k] >= 0 ? p++ : -1;
        }
    }
    if (tid == nthr - 1) {
        const int live = base + incl;
        const int tiles = (live + VS_BM_ROWS - 1) / VS_BM_ROWS;
        int splits = tiles > 0 ? items / tiles : 1;
        if (splits > max_splits) splits = max_splits;
        if (splits > vocab_tiles) splits = vocab_tiles;
        if (splits < 1) splits = 1;
        ctl[0] = live; ctl[1] = tiles; ctl[2] = splits; ctl[3] = (vocab_tiles + splits - 1) / splits;
    }
}
