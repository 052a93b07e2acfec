// Loss scalars of a step (shared by c1v_loss_reduce and the extra block of c1v_grad_finalize_loss).
#pragma once
#include "c1v_common.h"

// loss scalars (single block, deterministic order)
// (one workgroup of any whole number of waves <= 16; also run as an extra block of grad_finalize)
__device__ __forceinline__ void loss_reduce_body(const float* __restrict__ xent_rows, const int* __restrict__ ids,
                                                 const float* __restrict__ kl_b, int B, int T_len, int B_global,
                                                 float* __restrict__ out3, float* __restrict__ xent_b, const bool pre = false) {
    // pre: xent_b already holds the per-sample means (c1v_loss_sample_means: large batches -- 270 k rows at C4 kept this
    // single block busy for 70 us, twice as long as the rest of grad_finalize); only the sums over the samples are left.
    // Thread i owns sample i: it reads the T per-step values and ids of its row with independent 16-byte loads (all in
    // flight together: one memory round trip per sample, no LDS staging, no barrier) and sums them in step order.  Every
    // sum has a fixed order.
    __shared__ float sx[16], sk[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
    float ax = 0.f, ak = 0.f;
    constexpr int LR_TMAX = 128;                   // longest row a single thread sums
    if (pre) {
        for (int b = tid; b < B && tid < 256; b += 256) { ax += xent_b[b]; ak += kl_b ? kl_b[b] : 0.f; }
        ax = wave_sum(ax); ak = wave_sum(ak);
    } else if (T_len <= LR_TMAX) {
        const bool vec = (T_len & 3) == 0 && (((((uintptr_t)xent_rows) | ((uintptr_t)ids)) & 15) == 0);
        // samples go to the first 256 threads whatever the block size, so the order of every sum (and the result, bit for
        // bit) is the same in the stand-alone kernel and as a block of grad_finalize
        for (int b = tid; b < B && tid < 256; b += 256) {
            const float kl = kl_b ? kl_b[b] : 0.f;
            const float* xr = xent_rows + (int64_t)b * T_len;
            const int* ir = ids + (int64_t)b * T_len;
            float s = 0.f, c = 0.f;
            if (vec) {
#pragma unroll 8
                for (int t = 0; t < T_len; t += 4) {
                    const float4 xv = *(const float4*)(xr + t);
                    const int4 iv = *(const int4*)(ir + t);
                    s += iv.x >= 0 ? xv.x : 0.f; s += iv.y >= 0 ? xv.y : 0.f; s += iv.z >= 0 ? xv.z : 0.f; s += iv.w >= 0 ? xv.w : 0.f;
                    c += (iv.x >= 0 ? 1.f : 0.f); c += (iv.y >= 0 ? 1.f : 0.f); c += (iv.z >= 0 ? 1.f : 0.f); c += (iv.w >= 0 ? 1.f : 0.f);
                }
            } else {
#pragma unroll 8
                for (int t = 0; t < T_len; ++t) { s += ir[t] >= 0 ? xr[t] : 0.f; c += ir[t] >= 0 ? 1.f : 0.f; }
            }
            const float xb = s / c;
            if (xent_b) xent_b[b] = xb;
            ax += xb;
            ak += kl;
        }
        ax = wave_sum(ax); ak = wave_sum(ak);
    } else {
        // very long sequences: one wave per sample, lanes over the steps
        for (int b = wave; b < B; b += nwave) {
            float s = 0.f, c = 0.f;
            for (int t = lane; t < T_len; t += 64) {
                const bool on = ids[(int64_t)b * T_len + t] >= 0;
                s += on ? xent_rows[(int64_t)b * T_len + t] : 0.f; c += on ? 1.f : 0.f;
            }
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

// per-sample means for loss_reduce_body(pre = true): one wave per sample, lanes over the steps, fixed order
__device__ __forceinline__ void loss_sample_means_body(const float* __restrict__ xent_rows, const int* __restrict__ ids, int B, int T_len,
                                                       float* __restrict__ xent_b, int b) {
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f, c = 0.f;
    for (int t = lane; t < T_len; t += 64) {
        const bool on = ids[(int64_t)b * T_len + t] >= 0;
        s += on ? xent_rows[(int64_t)b * T_len + t] : 0.f; c += on ? 1.f : 0.f;
    }
    s = wave_sum_dpp(s); c = wave_sum_dpp(c);
    if (lane == 0) xent_b[b] = s / c;
}

constexpr int VS_BM_ROWS = 128;   // sample rows per workgroup of the vocabulary stream kernel
constexpr int VS_CUS = 256;       // compute units of an MI355X: workgroups of the stream kernel that run at the same time

// positions of the rows that have a target (ids[m] >= 0) in compact order, -1 for the others, and the work split of the
// vocabulary stream over them:
//   ctl[0] = n_live, ctl[1] = live row tiles (128 rows), ctl[2] = vocabulary splits per TAIL row tile, ctl[3] = tiles per split,
//   ctl[4] = FULL row tiles: the first ctl[4] row tiles (a multiple of VS_CUS) take the whole vocabulary in one workgroup each
// The stream kernels are launched with a fixed grid of `items` workgroups (a captured graph cannot resize it) and run one
// workgroup per CU.  Whole rounds of VS_CUS row tiles need no splitting; the remainder of the row tiles -- all of them, when
// there are fewer than VS_CUS -- is cut into as many vocabulary splits as make up one more round, so that the last round
// is as short as its share of the work (1051 row tiles: 4 rounds + 27 tiles x 9 splits, instead of a fifth whole round
// for 27 workgroups).
// One workgroup of any whole number of waves <= 16 (also run as a block of c1v_step_prologue).  Every WAVE owns a contiguous
// run of rows and walks it 256 rows at a time -- lane l holds rows 4l .. 4l+3 of the step as one 16-byte word, so loads and
// stores are whole 1 KB lines -- eight steps per round of requests; the order inside a step comes from a wave scan, the
// waves' bases from LDS.  (Round 1 gave every THREAD a contiguous run: 64 lanes x 16 bytes 1 KB apart per request and one
// request per round trip -- at C4, 262 k rows, 0.19 ms for this single block; 43 us now.  A one-pass form that keeps the
// flags of 128 steps in registers was 36 us alone but its unrolled code cost the C2 prologue 7 us: not kept.)
// rows / cids (optional): the inverse table -- rows[p] = m and cids[p] = ids[m] for the live row with position p -- so that
// the finish step can run over the compacted rows only (c1v_xent_finish_args::live_rows)
__device__ __forceinline__ void live_put4(int* __restrict__ rows, int* __restrict__ cids, const int4& v, const int4& o4, int m) {
    if (rows == nullptr) return;
    if (o4.x >= 0) { rows[o4.x] = m; cids[o4.x] = v.x; }
    if (o4.y >= 0) { rows[o4.y] = m + 1; cids[o4.y] = v.y; }
    if (o4.z >= 0) { rows[o4.z] = m + 2; cids[o4.z] = v.z; }
    if (o4.w >= 0) { rows[o4.w] = m + 3; cids[o4.w] = v.w; }
}
__device__ __forceinline__ void live_rows_body(const int* __restrict__ ids, int M, int items, int vocab_tiles, int max_splits,
                                               int* __restrict__ pos, int* __restrict__ ctl, int* __restrict__ rows = nullptr,
                                               int* __restrict__ cids = nullptr) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
    constexpr int UN = 8;
    const int per_wave = (((M + nwave - 1) / nwave + 255) / 256) * 256;       // rows per wave: whole 256-row steps
    const int w0 = wave * per_wave;
    const bool vec = ((((uintptr_t)ids) | ((uintptr_t)pos)) & 15) == 0;
    auto load4 = [&](int m) -> int4 {                                          // ids of rows m .. m+3 (-1 behind the end)
        if (vec && m + 3 < M) return *(const int4*)(ids + m);
        int4 v;
        v.x = m < M ? ids[m] : -1; v.y = m + 1 < M ? ids[m + 1] : -1; v.z = m + 2 < M ? ids[m + 2] : -1; v.w = m + 3 < M ? ids[m + 3] : -1;
        return v;
    };
    int cnt = 0;
    for (int s0 = 0; s0 < per_wave && w0 + s0 < M; s0 += 256 * UN) {
        int4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = load4(w0 + s0 + 256 * u + 4 * lane);
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (s0 + 256 * u < per_wave) cnt += (v[u].x >= 0) + (v[u].y >= 0) + (v[u].z >= 0) + (v[u].w >= 0);
    }
    int incl = cnt;                                    // the wave's total, in every lane
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) incl += __shfl_xor(incl, o, 64);
    if (lane == 0) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    int run = base;                                    // compact index of the wave's next live row
    for (int s0 = 0; s0 < per_wave && w0 + s0 < M; s0 += 256 * UN) {
        int4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = load4(w0 + s0 + 256 * u + 4 * lane);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (s0 + 256 * u >= per_wave) continue;
            const int m = w0 + s0 + 256 * u + 4 * lane;
            const int c = (v[u].x >= 0) + (v[u].y >= 0) + (v[u].z >= 0) + (v[u].w >= 0);
            int sc = c;                                // inclusive scan over the lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(sc, o, 64); if (lane >= o) sc += t; }
            int p = run + sc - c;
            run += __shfl(sc, 63, 64);
            int4 o4;
            o4.x = v[u].x >= 0 ? p++ : -1; o4.y = v[u].y >= 0 ? p++ : -1; o4.z = v[u].z >= 0 ? p++ : -1; o4.w = v[u].w >= 0 ? p++ : -1;
            live_put4(rows, cids, v[u], o4, m);
            if (vec && m + 3 < M) *(int4*)(pos + m) = o4;
            else {
                if (m < M) pos[m] = o4.x;
                if (m + 1 < M) pos[m + 1] = o4.y;
                if (m + 2 < M) pos[m + 2] = o4.z;
                if (m + 3 < M) pos[m + 3] = o4.w;
            }
        }
    }
    if (tid == nthr - 1) {
        const int live = base + incl;          // (last wave: its base + its total)
        const int tiles = (live + VS_BM_ROWS - 1) / VS_BM_ROWS;
        int full = (tiles / VS_CUS) * VS_CUS;
        if (full > items) full = (items / VS_CUS) * VS_CUS;   // (cannot happen with a grid that covers all row tiles)
        const int tail = tiles - full;
        int splits = tail > 0 ? (full > 0 ? VS_CUS : items) / tail : 1;
        if (tail > 0 && splits > (items - full) / tail) splits = (items - full) / tail;
        if (splits > max_splits) splits = max_splits;
        if (splits > vocab_tiles) splits = vocab_tiles;
        if (splits < 1) splits = 1;
        ctl[0] = live; ctl[1] = tiles; ctl[2] = splits; ctl[3] = (vocab_tiles + splits - 1) / splits; ctl[4] = full;
    }
}

// The same table from SEVERAL workgroups of one launch, none waiting for another (small batches: the single workgroup above
// walks 8 192 rows in 8 dependent 256-row steps per wave -- it was the longest job of the C2 step prologue).  Workgroup
// `blk` of `nblk` owns rows [blk * R, (blk + 1) * R), R = 256 rows per wave: ONE scan step per wave.  What lies in front of
// its range it counts itself (at most 15 R rows, read as 16-byte words, all requests in flight together) instead of
// waiting for the workgroups before it; the last workgroup also writes the control block.  Same results as
// live_rows_body, bit for bit (positions are integers).
constexpr int LIVE_MB_MAX = 16;        // workgroups at most (beyond that the redundant counting outgrows what it saves)
__device__ __forceinline__ void live_rows_mb_body(const int* __restrict__ ids, int M, int items, int vocab_tiles, int max_splits,
                                                  int* __restrict__ pos, int* __restrict__ ctl, int blk, int nblk,
                                                  int* __restrict__ rows = nullptr, int* __restrict__ cids = nullptr) {
    __shared__ int wcnt[16], wbef[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
    const int R = nwave * 256;
    const int r0 = blk * R;
    const bool vec = ((((uintptr_t)ids) | ((uintptr_t)pos)) & 15) == 0;
    auto load4 = [&](int m) -> int4 {                                          // ids of rows m .. m+3 (-1 behind the end)
        if (vec && m + 3 < M) return *(const int4*)(ids + m);
        int4 v;
        v.x = m < M ? ids[m] : -1; v.y = m + 1 < M ? ids[m + 1] : -1; v.z = m + 2 < M ? ids[m + 2] : -1; v.w = m + 3 < M ? ids[m + 3] : -1;
        return v;
    };
    // this wave's step and the rows in front of the workgroup's range, requested together
    const int m = r0 + wave * 256 + 4 * lane;
    const int4 own = load4(m);
    int before = 0;
    constexpr int PU = 8;                              // requests per thread and round of the prefix count
    for (int q0 = 4 * tid; q0 < r0; q0 += 4 * nthr * PU) {
        int4 v[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) { const int q = q0 + u * 4 * nthr; v[u] = q < r0 ? load4(q) : make_int4(-1, -1, -1, -1); }
#pragma unroll
        for (int u = 0; u < PU; ++u) before += (v[u].x >= 0) + (v[u].y >= 0) + (v[u].z >= 0) + (v[u].w >= 0);
    }
    const int c = (own.x >= 0) + (own.y >= 0) + (own.z >= 0) + (own.w >= 0);
    int sc = c;                                        // inclusive scan of the step over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(sc, o, 64); if (lane >= o) sc += t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 63) wcnt[wave] = sc;                   // the step's total
    if (lane == 0) wbef[wave] = before;                // this wave's share of the rows in front of the range
    __syncthreads();
    int base = 0;
    for (int w = 0; w < nwave; ++w) base += wbef[w];
    for (int w = 0; w < wave; ++w) base += wcnt[w];
    int p = base + sc - c;
    int4 o4;
    o4.x = own.x >= 0 ? p++ : -1; o4.y = own.y >= 0 ? p++ : -1; o4.z = own.z >= 0 ? p++ : -1; o4.w = own.w >= 0 ? p++ : -1;
    live_put4(rows, cids, own, o4, m);
    if (vec && m + 3 < M) *(int4*)(pos + m) = o4;
    else {
        if (m < M) pos[m] = o4.x;
        if (m + 1 < M) pos[m + 1] = o4.y;
        if (m + 2 < M) pos[m + 2] = o4.z;
        if (m + 3 < M) pos[m + 3] = o4.w;
    }
    if (blk == nblk - 1 && tid == nthr - 1) {
        const int live = base + sc;                    // (last lane of the last wave: its inclusive scan ends the table)
        const int tiles = (live + VS_BM_ROWS - 1) / VS_BM_ROWS;
        int full = (tiles / VS_CUS) * VS_CUS;
        if (full > items) full = (items / VS_CUS) * VS_CUS;
        const int tail = tiles - full;
        int splits = tail > 0 ? (full > 0 ? VS_CUS : items) / tail : 1;
        if (tail > 0 && splits > (items - full) / tail) splits = (items - full) / tail;
        if (splits > max_splits) splits = max_splits;
        if (splits > vocab_tiles) splits = vocab_tiles;
        if (splits < 1) splits = 1;
        ctl[0] = live; ctl[1] = tiles; ctl[2] = splits; ctl[3] = (vocab_tiles + splits - 1) / splits; ctl[4] = full;
    }
}

