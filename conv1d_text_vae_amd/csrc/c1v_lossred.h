// Loss scalars of a step (shared by c1v_loss_reduce and the extra block of c1v_grad_finalize_loss).
#pragma once
#include "c1v_common.h"

// loss scalars (single block, deterministic order)
constexpr int LR_CAP = 4096;                       // row values staged per pass
// (one workgroup of any whole number of waves <= 16; also run as an extra block of grad_finalize)
__device__ __forceinline__ void loss_reduce_body(const float* __restrict__ xent_rows, const int* __restrict__ ids,
                                                 const float* __restrict__ kl_b, int B, int T_len, int B_global,
                                                 float* __restrict__ out3, float* __restrict__ xent_b) {
    // Pass structure: all threads stage the per-step values of S = LR_CAP / (T + 1) samples in LDS with independent,
    // coalesced loads (a wave-per-sample loop serialised one global-load latency per sample: 16 us at B = 256), then
    // thread i sums the T steps of sample i in step order (pitch T + 1: conflict-free).  Every sum has a fixed order.
    __shared__ float vx[LR_CAP], vc[LR_CAP];
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
#pragma unroll 4
            for (int i = tid; i < n; i += nthr) {
                int bl = (int)(((float)i + 0.5f) * rcp);
                int t = i - bl * T_len;
                if (t < 0) { --bl; t += T_len; } else if (t >= T_len) { ++bl; t -= T_len; }
                vx[bl * pitch + t] = xr[i];
                vc[bl * pitch + t] = ir[i] >= 0 ? 1.f : 0.f;
            }
            __syncthreads();
            for (int bl = tid; bl < ns; bl += nthr) {
                float s = 0.f, c = 0.f;
                for (int t = 0; t < T_len; ++t) { s += vx[bl * pitch + t]; c += vc[bl * pitch + t]; }
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

