// MFMA GEMM engine of libc1dvae (gfx950): every GEMM-shaped op of the VAE goes through one of two kernels.
//
//   gemm_nt : C[m][n] = sum_k A[m][k] * W[n][k]      conv / deconv / dense forward and dgrad (implicit GEMM:
//             a conv window over a channels-last buffer IS a contiguous K = k*cp run, so conv = GEMM with
//             overlapping A rows, lda = cp), with the fused epilogue of c1v_epilogue.
//   gemm_tn : C[i][j] = sum_r A[r][i] * D[r][j]      wgrad, split over r into fp32 slabs; optionally also the
//             column sums of D (bias gradient) through one extra all-ones MFMA operand.
//
// Both are 256 threads = 4 waves (2x2) and written once for "16 bytes of K per lane": bf16 feeds one
// v_mfma_f32_16x16x32_bf16 per 16-byte fragment pair, fp32 feeds four v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) --
// the k permutation inside a fragment is the same for both operands, so the sum is unchanged.
//
// Staging: operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging) into a ring of NS
// stages; a counted s_waitcnt vmcnt leaves NS-2 tiles in flight across the single barrier per K step.  At the shapes
// of this model (M = batch*seq of a few thousand rows, K a few hundred) the GEMMs are latency- not MFMA-bound, so the
// depth of the ring, not the MFMA schedule, is what sets their time.  LDS-DMA writes are lane-linear, therefore
// swizzles are applied to the per-lane SOURCE address and padding lanes read a dummy address.
// No load is predicated: out-of-range rows are clamped (their results are masked in the epilogue) and the K / row
// tails rely on zero padding of the operand buffers (documented in include/c1dvae.h).
#pragma once
#include "c1v_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------------
// fragment MMA: 16 bytes of K per lane
// ------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void mma16(f32x4_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<bf16_raw>(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) { c1v_glds16(src, c1v_lds_addr(lds_wave_base)); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void block_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// LDS image of an NT operand tile: 128-byte rows (64 bf16 / 32 fp32 of K), 16-byte chunk index XOR-swizzled with
// (row>>1)&7 so that the 16 lanes of a ds_read_b128 group (rows r..r+15, one chunk column) hit 16 distinct slots.
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// ------------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------
// fused epilogue (c1v_epilogue in include/c1dvae.h) over a tile parked in LDS
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store4(const EpiDev& e, int64_t idx, const float v[4], int count) {
    // `count` leading values are stored; one 8- / 16-byte store when all four are and the address allows it
    if (count == 4) {
        if (e.dst_dtype == C1V_F32) {
            float* d = (float*)e.dst + idx;
            if ((((uintptr_t)d) & 15) == 0) { *(float4*)d = make_float4(v[0], v[1], v[2], v[3]); return; }
        } else {
            bf16_raw* d = (bf16_raw*)e.dst + idx;
            if ((((uintptr_t)d) & 7) == 0) { *(uint2*)d = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); return; }
        }
    }
    for (int k = 0; k < count; ++k) store_any(e.dst, e.dst_dtype, idx + k, v[k]);
}

// A thread owns ONE 4-channel group (g = tid % G) of every (256/G)-th tile row, so its bias values are loaded once -- by
// the kernel, before the K loop, where the latency is free -- and its rows are processed in chunks of up to four: first
// the mask / dropout / pool bytes of the whole chunk are requested, then values are formed and stored (a rolled
// one-row-per-iteration loop serialised one global-load latency per row, ~1.5 us each).
template <int BM, int BN, int MODE>
__device__ __forceinline__ void epilogue_rows(const EpiDev& e, const float* ct, int LDC, int m0, int n0, int tid, const float bias4[4]) {
    constexpr bool PAIRS = MODE == C1V_ROW_POOL2 || MODE == C1V_ROW_PAIRSUM;
    constexpr int G = BN / 4, RPI = 256 / G, ROWS = PAIRS ? BM / 2 : BM, ITEMS = ROWS / RPI, CH = ITEMS < 4 ? ITEMS : 4;
    static_assert(256 % G == 0 && ROWS % RPI == 0 && ITEMS % CH == 0, "tile shape");
    const int g = tid % G, r0 = tid / G;
    const int n = n0 + 4 * g;
    if (n >= e.N) return;
    const int cnt = e.N - n < 4 ? e.N - n : 4;                      // valid channels of this group
    int tn = 0, c = n;
    if (e.reshape_w > 0) { tn = n / e.reshape_w; c = n - tn * e.reshape_w; }
    const bool straddle = e.reshape_w > 0 && c + cnt > e.reshape_w;  // group crosses a (t, c) row boundary: element-wise
    constexpr int mode = MODE;                        // compile-time: every mode test below folds away
#pragma unroll 1
    for (int j0 = 0; j0 < ITEMS; j0 += CH) {
        bool ok[CH];
        int bb[CH], uu[CH], tt[CH];
        float am[CH][4];
        uint8_t by[CH][4];
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int rr = r0 + (j0 + jj) * RPI;
            const int m = m0 + (PAIRS ? 2 * rr : rr);
            const int mm = m < e.M ? m : e.M - 1;
            const int b = mm / e.Le, u = mm - b * e.Le;
            const int t = e.reshape_w > 0 ? tn : (PAIRS ? (u >> 1) : u);
            bool o = m < e.M;
            if (mode == C1V_ROW_STORE) o = o && u < e.nrows && t < e.dst_len;
            else if (mode == C1V_ROW_UP2) o = o && u < e.nrows && 2 * t < e.dst_len;
            else if (mode == C1V_ROW_UNPOOL) o = o && u < e.nrows && 2 * t + 1 < e.dst_len;
            else if (mode == C1V_ROW_POOL2) o = o && t < e.dst_len && u + 1 < e.nrows;
            else o = o && t < e.dst_len && u < e.nrows;
            ok[jj] = o; bb[jj] = b; uu[jj] = u; tt[jj] = t;
#pragma unroll
            for (int k = 0; k < 4; ++k) { am[jj][k] = 1.0f; by[jj][k] = 1; }
            if (o && !straddle) {
                if (e.act && mode != C1V_ROW_UP2 && mode != C1V_ROW_POOL2) {
                    const int64_t ai = (int64_t)b * e.act_bstride + (int64_t)(e.act_off + t * e.act_rowmul) * e.act_cp + c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < cnt) am[jj][k] = load_any(e.act, e.act_dtype, ai + k);
                }
                const int64_t si = (int64_t)b * e.sel_bstride + (int64_t)t * e.sel_cp + c;
                if (mode == C1V_ROW_UNPOOL) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < cnt) by[jj][k] = e.sel[si + k];
                } else if (e.keep && (mode == C1V_ROW_STORE || mode == C1V_ROW_POOL2)) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k < cnt) by[jj][k] = e.keep[si + k];
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            if (!ok[jj]) continue;
            const int rr = r0 + (j0 + jj) * RPI;
            const int row = PAIRS ? 2 * rr : rr;
            const int b = bb[jj], u = uu[jj], t = tt[jj];
            const float4 q0 = *(const float4*)(ct + row * LDC + 4 * g);
            float v[4] = {q0.x, q0.y, q0.z, q0.w};
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            if (PAIRS) { const float4 q1 = *(const float4*)(ct + (row + 1) * LDC + 4 * g); w[0] = q1.x; w[1] = q1.y; w[2] = q1.z; w[3] = q1.w; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] += bias4[k]; w[k] += bias4[k];
                if (e.relu) { v[k] = fmaxf(v[k], 0.0f); w[k] = fmaxf(w[k], 0.0f); }
                if (e.exp2) v[k] = exp2f(v[k]);
                am[jj][k] = (e.act && mode != C1V_ROW_UP2 && mode != C1V_ROW_POOL2) ? (am[jj][k] > 0.0f ? e.act_scale : 0.0f) : 1.0f;
            }
            if (straddle) {
                // element-wise (t, c) = divmod(n + k, reshape_w); only dense layers whose width is not a multiple of 4
                for (int k = 0; k < cnt; ++k) {
                    const int te = (n + k) / e.reshape_w, ce = (n + k) - te * e.reshape_w;
                    float x = v[k];
                    if (mode == C1V_ROW_STORE) {
                        if (te >= e.dst_len) continue;
                        if (e.act) x *= load_any(e.act, e.act_dtype, (int64_t)b * e.act_bstride + (int64_t)(e.act_off + te * e.act_rowmul) * e.act_cp + ce) > 0.0f ? e.act_scale : 0.0f;
                        if (e.keep) x = e.keep[(int64_t)b * e.sel_bstride + (int64_t)te * e.sel_cp + ce] ? x * e.keep_scale : 0.0f;
                        store_any(e.dst, e.dst_dtype, (int64_t)b * e.dst_bstride + (int64_t)(e.dst_off + te) * e.dst_cp + ce, x);
                    } else {   // UNPOOL
                        if (2 * te + 1 >= e.dst_len) continue;
                        if (e.act) x *= load_any(e.act, e.act_dtype, (int64_t)b * e.act_bstride + (int64_t)(e.act_off + te * e.act_rowmul) * e.act_cp + ce) > 0.0f ? e.act_scale : 0.0f;
                        const int sidx = e.sel[(int64_t)b * e.sel_bstride + (int64_t)te * e.sel_cp + ce] ? 1 : 0;
                        const int64_t d = (int64_t)b * e.dst_bstride + (int64_t)(e.dst_off + 2 * te) * e.dst_cp + ce;
                        store_any(e.dst, e.dst_dtype, d + (int64_t)sidx * e.dst_cp, x);
                        store_any(e.dst, e.dst_dtype, d + (int64_t)(1 - sidx) * e.dst_cp, 0.0f);
                    }
                }
                continue;
            }
            const int64_t si = (int64_t)b * e.sel_bstride + (int64_t)t * e.sel_cp + c;
            if (mode == C1V_ROW_STORE) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] *= am[jj][k];
                    if (e.keep) v[k] = by[jj][k] ? v[k] * e.keep_scale : 0.0f;
                }
                store4(e, (int64_t)b * e.dst_bstride + (int64_t)(e.dst_off + t) * e.dst_cp + c, v, cnt);
            } else if (mode == C1V_ROW_UP2) {
                const int64_t d = (int64_t)b * e.dst_bstride + (int64_t)(e.dst_off + 2 * t) * e.dst_cp + c;
                store4(e, d, v, cnt);
                if (2 * t + 1 < e.dst_len) store4(e, d + e.dst_cp, v, cnt);
            } else if (mode == C1V_ROW_UNPOOL) {
                float hi[4], lo[4];                                  // row 2t and row 2t+1
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool s1 = k < cnt && by[jj][k] != 0;
                    const float x = v[k] * am[jj][k];
                    hi[k] = s1 ? 0.0f : x; lo[k] = s1 ? x : 0.0f;
                }
                const int64_t d = (int64_t)b * e.dst_bstride + (int64_t)(e.dst_off + 2 * t) * e.dst_cp + c;
                store4(e, d, hi, cnt);
                store4(e, d + e.dst_cp, lo, cnt);
            } else if (mode == C1V_ROW_POOL2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool s1 = w[k] > v[k];                    // first maximum wins ties
                    float x = s1 ? w[k] : v[k];
                    if (k < cnt) {
                        if (e.sel) e.sel[si + k] = s1 ? 1 : 0;
                        if (e.keep) x = by[jj][k] ? x * e.keep_scale : 0.0f;
                    }
                    v[k] = x;
                }
                store4(e, (int64_t)b * e.dst_bstride + (int64_t)(e.dst_off + t) * e.dst_cp + c, v, cnt);
            } else {   // PAIRSUM
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ((u < e.nrows ? v[k] : 0.0f) + (u + 1 < e.nrows ? w[k] : 0.0f)) * am[jj][k];
                store4(e, (int64_t)b * e.dst_bstride + (int64_t)(e.dst_off + t) * e.dst_cp + c, v, cnt);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// vectorised epilogue (EpiDev::fastw = 4 or 2, decided on the host by c1v_fill_epi): every address a 4-channel group
// touches is a multiple of VW elements, so activations / mask bytes / results move as whole 4-, 8- or 16-byte words and
// nothing is branched per element.  The inputs of the first chunk of rows are requested BEFORE the K loop (their
// latency hides behind it); the row -> (sample, step) split uses one scalar division per workgroup (RowDiv).
// ------------------------------------------------------------------------------------------------------
struct RowDiv { int Le, b0, u0; float rcp; };
__device__ __forceinline__ RowDiv make_rowdiv(int m0, int Le) {
    RowDiv d;
    d.Le = Le; d.b0 = m0 / Le; d.u0 = m0 - d.b0 * Le; d.rcp = 1.0f / (float)Le;
    return d;
}
// (b, u) of row m0 + r, 0 <= r < 2^15: float estimate of the quotient, exact after one correction step
__device__ __forceinline__ void rowdiv(const RowDiv& d, int r, int& b, int& u) {
    const int x = d.u0 + r;
    int q = (int)((float)x * d.rcp);
    int rem = x - q * d.Le;
    if (rem < 0) { --q; rem += d.Le; } else if (rem >= d.Le) { ++q; rem -= d.Le; }
    b = d.b0 + q; u = rem;
}

struct EpiState {
    static constexpr int CH = 4;
    int64_t di[CH];        // destination element index of the group (row 2t for UP2 / UNPOOL)
    int si[CH];            // index of the group's mask bytes (sel / keep)
    uint4 act[CH];         // raw activation words: bf16 in .x .y, fp32 in .x .. .w   (LATENT_BWD: z_mean)
    uint4 x1[CH], x2[CH];  // LATENT_BWD only: z_log_var and eps of the group
    uint32_t by[CH];       // four mask bytes
    uint32_t flags;        // 4 bits per row: 1 = stored, 2 = UP2: second row fits / PAIRSUM: first member valid, 4 = PAIRSUM: second member valid
    int cnt;               // valid channels of the group (VW = 4: 4, or 1..3 in a padded row's last group; VW = 2: 2 or 4); 0 = thread idle
};

template <int VW> __device__ __forceinline__ uint4 epi_ld_act(const EpiDev& e, int64_t i, int cnt) {
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (e.act_dtype == C1V_F32) {
        const float* p = (const float*)e.act + i;
        if (VW == 4) r = *(const uint4*)p;
        else { const uint2 a = *(const uint2*)p; r.x = a.x; r.y = a.y; if (cnt == 4) { const uint2 b = *(const uint2*)(p + 2); r.z = b.x; r.w = b.y; } }
    } else {
        const bf16_raw* p = (const bf16_raw*)e.act + i;
        if (VW == 4) { const uint2 a = *(const uint2*)p; r.x = a.x; r.y = a.y; }
        else { r.x = *(const uint32_t*)p; if (cnt == 4) r.y = *(const uint32_t*)(p + 2); }
    }
    return r;
}
template <int VW> __device__ __forceinline__ uint32_t epi_ld_bytes(const uint8_t* p, int cnt) {
    if (VW == 4) return *(const uint32_t*)p;
    uint32_t r = *(const uint16_t*)p;
    r |= (cnt == 4 ? (uint32_t)*(const uint16_t*)(p + 2) : 0x0101u) << 16;
    return r;
}
template <int VW> __device__ __forceinline__ void epi_st_bytes(uint8_t* p, uint32_t v, int cnt) {
    if (VW == 4) { *(uint32_t*)p = v; return; }
    *(uint16_t*)p = (uint16_t)v;
    if (cnt == 4) *(uint16_t*)(p + 2) = (uint16_t)(v >> 16);
}
template <int VW> __device__ __forceinline__ void epi_st(const EpiDev& e, int64_t i, const float v[4], int cnt) {
    if (e.dst_dtype == C1V_F32) {
        float* d = (float*)e.dst + i;
        if (VW == 4) *(float4*)d = make_float4(v[0], v[1], v[2], v[3]);
        else { *(float2*)d = make_float2(v[0], v[1]); if (cnt == 4) *(float2*)(d + 2) = make_float2(v[2], v[3]); }
    } else {
        bf16_raw* d = (bf16_raw*)e.dst + i;
        if (VW == 4) *(uint2*)d = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        else { *(uint32_t*)d = pack_bf16x2(v[0], v[1]); if (cnt == 4) *(uint32_t*)(d + 2) = pack_bf16x2(v[2], v[3]); }
    }
}

template <int BM, int BN, int MODE> struct EpiShape {
    static constexpr bool PAIRS = MODE == C1V_ROW_POOL2 || MODE == C1V_ROW_PAIRSUM;
    static constexpr int G = BN / 4, RPI = 256 / G, ROWS = PAIRS ? BM / 2 : BM, ITEMS = ROWS / RPI, CH = ITEMS < 4 ? ITEMS : 4;
    static_assert(256 % G == 0 && ROWS % RPI == 0 && ITEMS % CH == 0 && (ITEMS == CH || ITEMS % (2 * CH) == 0), "tile shape");
};

template <int BM, int BN, int MODE, int VW>
__device__ __forceinline__ void epi_load(const EpiDev& e, const RowDiv& rd, EpiState& st, int m0, int n0, int tid, int j0) {
    using S = EpiShape<BM, BN, MODE>;
    constexpr bool PAIRS = S::PAIRS;
    constexpr int G = S::G, RPI = S::RPI, CH = S::CH;
    constexpr bool USE_ACT = MODE != C1V_ROW_UP2 && MODE != C1V_ROW_POOL2 && MODE != C1V_ROW_LATENT_BWD;
    const int g = tid % G, r0 = tid / G;
    const int n = n0 + 4 * g;
    st.flags = 0; st.cnt = 0;
    if (n >= e.N) return;
    // (VW = 4 with a partial last group -- c1v_fill_epi's pad_tail: the words are whole, cnt tells STORE which values to zero)
    const int cnt = VW == 4 ? (e.N - n >= 4 ? 4 : e.N - n) : (e.N - n >= 4 ? 4 : 2);
    st.cnt = cnt;
    int tn = 0, c = n;
    if (e.reshape_w > 0) { tn = n / e.reshape_w; c = n - tn * e.reshape_w; }
#pragma unroll
    for (int jj = 0; jj < CH; ++jj) {
        const int rr = r0 + (j0 + jj) * RPI;
        const int r = PAIRS ? 2 * rr : rr;
        int b, u;
        rowdiv(rd, r, b, u);
        const int t = e.reshape_w > 0 ? tn : (PAIRS ? (u >> 1) : u);
        bool o = m0 + r < e.M;
        uint32_t f = 0;
        if (MODE == C1V_ROW_STORE || MODE == C1V_ROW_LATENT_BWD) o = o && u < e.nrows && t < e.dst_len;
        else if (MODE == C1V_ROW_UP2) { o = o && u < e.nrows && 2 * t < e.dst_len; f = 2 * t + 1 < e.dst_len ? 2u : 0u; }
        else if (MODE == C1V_ROW_UNPOOL) o = o && u < e.nrows && 2 * t + 1 < e.dst_len;
        else if (MODE == C1V_ROW_POOL2) o = o && t < e.dst_len && u + 1 < e.nrows;
        else { o = o && t < e.dst_len && u < e.nrows; f = (u < e.nrows ? 2u : 0u) | (u + 1 < e.nrows ? 4u : 0u); }
        const int trow = (MODE == C1V_ROW_UP2 || MODE == C1V_ROW_UNPOOL) ? 2 * t : t;
        st.di[jj] = (int64_t)b * e.dst_bstride + (int64_t)(e.dst_off + trow) * e.dst_cp + c;
        st.si[jj] = (int)((int64_t)b * e.sel_bstride + (int64_t)t * e.sel_cp + c);
        st.act[jj] = make_uint4(0u, 0u, 0u, 0u);
        st.by[jj] = 0x01010101u;
        if (MODE == C1V_ROW_LATENT_BWD) {
            st.x1[jj] = st.act[jj]; st.x2[jj] = st.act[jj];
            if (o) {                                      // VW == 4: whole 16-byte words of z_mean, z_log_var, eps; 4 mask bytes
                const int64_t row = (int64_t)b * e.lat_D + c;
                st.act[jj] = *(const uint4*)(e.lat_mlv + 2 * (int64_t)b * e.lat_D + c);
                st.x1[jj] = *(const uint4*)(e.lat_mlv + 2 * (int64_t)b * e.lat_D + e.lat_D + c);
                st.x2[jj] = *(const uint4*)(e.lat_eps + row);
                if (e.lat_keep) st.by[jj] = *(const uint32_t*)(e.lat_keep + row);
            }
        }
        if (o) {
            st.flags |= (1u | f) << (4 * jj);
            if (USE_ACT && e.act)
                st.act[jj] = epi_ld_act<VW>(e, (int64_t)b * e.act_bstride + (int64_t)(e.act_off + t * e.act_rowmul) * e.act_cp + c, cnt);
            if (MODE == C1V_ROW_UNPOOL) st.by[jj] = epi_ld_bytes<VW>(e.sel + st.si[jj], cnt);
            else if (e.keep && (MODE == C1V_ROW_STORE || MODE == C1V_ROW_POOL2)) st.by[jj] = epi_ld_bytes<VW>(e.keep + st.si[jj], cnt);
        }
    }
}

template <int BM, int BN, int MODE, int VW>
__device__ __forceinline__ void epi_finish(const EpiDev& e, const EpiState& st, const float* ct, int LDC, int tid, int j0, const float bias4[4]) {
    using S = EpiShape<BM, BN, MODE>;
    constexpr bool PAIRS = S::PAIRS;
    constexpr int G = S::G, RPI = S::RPI, CH = S::CH;
    constexpr bool USE_ACT = MODE != C1V_ROW_UP2 && MODE != C1V_ROW_POOL2 && MODE != C1V_ROW_LATENT_BWD;
    if (st.flags == 0) return;
    const int g = tid % G, r0 = tid / G;
    const int cnt = st.cnt;
    const bool have_act = USE_ACT && e.act != nullptr;
#pragma unroll
    for (int jj = 0; jj < CH; ++jj) {
        const uint32_t f = (st.flags >> (4 * jj)) & 15u;
        if (!(f & 1u)) continue;
        const int rr = r0 + (j0 + jj) * RPI;
        const int row = PAIRS ? 2 * rr : rr;
        const float4 q0 = *(const float4*)(ct + row * LDC + 4 * g);
        float v[4] = {q0.x, q0.y, q0.z, q0.w};
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (PAIRS) { const float4 q1 = *(const float4*)(ct + (row + 1) * LDC + 4 * g); w[0] = q1.x; w[1] = q1.y; w[2] = q1.z; w[3] = q1.w; }
        float am[4];
        const uint32_t aw[4] = {st.act[jj].x, st.act[jj].y, st.act[jj].z, st.act[jj].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] += bias4[k]; w[k] += bias4[k];
            if (e.relu) { v[k] = fmaxf(v[k], 0.0f); w[k] = fmaxf(w[k], 0.0f); }
            if (e.exp2) v[k] = exp2f(v[k]);
            float a = 1.0f;
            if (have_act) {
                const float x = e.act_dtype == C1V_F32 ? __uint_as_float(aw[k]) : __uint_as_float((k & 1) ? (aw[k >> 1] & 0xFFFF0000u) : (aw[k >> 1] << 16));
                a = x > 0.0f ? e.act_scale : 0.0f;
            }
            am[k] = a;
        }
        const uint32_t by = st.by[jj];
        const int64_t d = st.di[jj];
        if (MODE == C1V_ROW_LATENT_BWD) {
            // c1v_latent_head_bwd on the fly: v = d loss / d zd
            const uint32_t mw[4] = {st.act[jj].x, st.act[jj].y, st.act[jj].z, st.act[jj].w};
            const uint32_t lw[4] = {st.x1[jj].x, st.x1[jj].y, st.x1[jj].z, st.x1[jj].w};
            const uint32_t ew[4] = {st.x2[jj].x, st.x2[jj].y, st.x2[jj].z, st.x2[jj].w};
            float dm[4], dl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float dz = v[k];
                if (e.lat_keep) dz = ((by >> (8 * k)) & 255u) ? dz / e.lat_keep_prob : 0.0f;
                const float mv = __uint_as_float(mw[k]), l = __uint_as_float(lw[k]);
                dm[k] = dz + 2.0f * mv * e.lat_kc;
                dl[k] = dz * __uint_as_float(ew[k]) * e.lat_std_scale * expf(e.lat_std_scale * l) + (expf(l) - 1.0f) * e.lat_kc;
            }
            epi_st<VW>(e, d, dm, cnt);
            epi_st<VW>(e, d + e.lat_D, dl, cnt);
        } else if (MODE == C1V_ROW_STORE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] *= am[k];
                if (e.keep) v[k] = ((by >> (8 * k)) & 255u) ? v[k] * e.keep_scale : 0.0f;
                if (VW == 4 && k >= cnt) v[k] = 0.0f;           // row padding behind channel N stays zero
            }
            epi_st<VW>(e, d, v, cnt);
        } else if (MODE == C1V_ROW_UP2) {
            epi_st<VW>(e, d, v, cnt);
            if (f & 2u) epi_st<VW>(e, d + e.dst_cp, v, cnt);
        } else if (MODE == C1V_ROW_UNPOOL) {
            float hi[4], lo[4];                                  // row 2t and row 2t+1
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool s1 = ((by >> (8 * k)) & 255u) != 0;
                const float x = v[k] * am[k];
                hi[k] = s1 ? 0.0f : x; lo[k] = s1 ? x : 0.0f;
            }
            epi_st<VW>(e, d, hi, cnt);
            epi_st<VW>(e, d + e.dst_cp, lo, cnt);
        } else if (MODE == C1V_ROW_POOL2) {
            uint32_t sb = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool s1 = w[k] > v[k];                    // first maximum wins ties
                float x = s1 ? w[k] : v[k];
                sb |= (s1 ? 1u : 0u) << (8 * k);
                if (e.keep) x = ((by >> (8 * k)) & 255u) ? x * e.keep_scale : 0.0f;
                v[k] = x;
            }
            if (e.sel) epi_st_bytes<VW>(e.sel + st.si[jj], sb, cnt);
            epi_st<VW>(e, d, v, cnt);
        } else {   // PAIRSUM
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (((f & 2u) ? v[k] : 0.0f) + ((f & 4u) ? w[k] : 0.0f)) * am[k];
            epi_st<VW>(e, d, v, cnt);
        }
    }
}

// all rows of the tile: chunk 0 was requested before the K loop (s0); further chunks (128-row tiles) ping-pong
template <int BM, int BN, int MODE, int VW>
__device__ __forceinline__ void epi_fast_rows(const EpiDev& e, const RowDiv& rd, EpiState& s0, const float* ct, int LDC, int m0, int n0, int tid,
                                              const float bias4[4]) {
    using S = EpiShape<BM, BN, MODE>;
    if constexpr (S::ITEMS == S::CH) {
        epi_finish<BM, BN, MODE, VW>(e, s0, ct, LDC, tid, 0, bias4);
    } else {
        EpiState s1;
#pragma unroll 1
        for (int j0 = 0; j0 < S::ITEMS; j0 += 2 * S::CH) {
            epi_load<BM, BN, MODE, VW>(e, rd, s1, m0, n0, tid, j0 + S::CH);
            epi_finish<BM, BN, MODE, VW>(e, s0, ct, LDC, tid, j0, bias4);
            if (j0 + 2 * S::CH < S::ITEMS) epi_load<BM, BN, MODE, VW>(e, rd, s0, m0, n0, tid, j0 + 2 * S::CH);
            epi_finish<BM, BN, MODE, VW>(e, s1, ct, LDC, tid, j0 + S::CH, bias4);
        }
    }
}

// the same rows with ONE state: every chunk is requested, then finished (large-GEMM variant: a second workgroup on the CU
// hides the latency, and two states in flight would not fit its register budget)
template <int BM, int BN, int MODE, int VW>
__device__ __forceinline__ void epi_seq_rows(const EpiDev& e, const RowDiv& rd, const float* ct, int LDC, int m0, int n0, int tid,
                                             const float bias4[4]) {
    using S = EpiShape<BM, BN, MODE>;
    if constexpr (MODE != C1V_ROW_LATENT_BWD) {          // (the latent head's dgrad is a small GEMM: never this variant)
        EpiState s;
#pragma unroll 1
        for (int j0 = 0; j0 < S::ITEMS; j0 += S::CH) {
            epi_load<BM, BN, MODE, VW>(e, rd, s, m0, n0, tid, j0);
            epi_finish<BM, BN, MODE, VW>(e, s, ct, LDC, tid, j0, bias4);
        }
    }
}

// wave-uniform dispatch over the row mode at a fixed vector width
#define C1V_EPI_DISPATCH(VWV, CALL)                                                      \
    switch (p.e.row_mode) {                                                              \
        case C1V_ROW_STORE:   { constexpr int MODE_ = C1V_ROW_STORE;   constexpr int VW_ = VWV; CALL; } break; \
        case C1V_ROW_POOL2:   { constexpr int MODE_ = C1V_ROW_POOL2;   constexpr int VW_ = VWV; CALL; } break; \
        case C1V_ROW_UP2:     { constexpr int MODE_ = C1V_ROW_UP2;     constexpr int VW_ = VWV; CALL; } break; \
        case C1V_ROW_PAIRSUM: { constexpr int MODE_ = C1V_ROW_PAIRSUM; constexpr int VW_ = VWV; CALL; } break; \
        case C1V_ROW_LATENT_BWD: { constexpr int MODE_ = C1V_ROW_LATENT_BWD; constexpr int VW_ = VWV; CALL; } break; \
        default:              { constexpr int MODE_ = C1V_ROW_UNPOOL;  constexpr int VW_ = VWV; CALL; } break; \
    }

#ifdef NT_STAMP
// diagnostic build only: cycle stamps of wave 0 of every workgroup (start, after prologue, after K loop, end)
__device__ unsigned long long nt_dbg[16 * 8192];
extern "C" int c1v_debug_read_nt(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(nt_dbg), sizeof(unsigned long long) * n);
}
#define NT_T(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i]) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#define NT_TICK(i) do { unsigned long long now_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    tacc[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define NT_T(i) do {} while (0)
#define NT_TICK(i) do {} while (0)
#endif

// Workgroup = 6 waves: waves 0-3 compute (2x2 over the tile), waves 4 and 5 only feed the LDS ring (wave 4 the A rows,
// wave 5 the W rows).  Measured with per-phase s_memtime stamps when all four waves did both jobs: the LDS-DMA requests
// of one 16 KiB K tile take ~270 cycles to ISSUE (the CU accepts 64 B/clk) and block the issuing wave meanwhile, serially
// with its ~170 cycles of fragment reads and ~200 of MFMA -- 580 cycles per tile against 256 for the data alone.
constexpr int C1V_GEMM_THREADS = 384;

// PRE: the epilogue inputs of the first chunk of rows are requested before the K loop (small GEMMs: their latency hides
// behind it).  The large-GEMM variant (PRE = false) asks for them after the K loop instead: they would pin ~66 registers
// through it, and that variant runs TWO workgroups per CU (<= 168 registers), where the other workgroup's K loop hides
// this one's epilogue.
// XCD-aware deal of a large GEMM's output tiles.  Workgroups go to the 8 XCDs round-robin in dispatch order and every XCD
// has an L2 of its own; with row tiles fastest in the grid the column tiles of one row tile ran thousands of workgroups
// apart, so every column tile fetched its activation rows from HBM again (2-4x the operand at 256-512 output channels).
// Here XCD k (= id % 8) owns a CONTIGUOUS range of the tiles in (row tile, column tile) order, column tile fastest: the
// workgroups that share an activation row tile are dispatched back to back to the same XCD and share its L2 copy.
__device__ __forceinline__ void xcd_tile(int id, int gm, int gn, int& bx, int& by) {
    const int total = gm * gn, k = id & 7, j = id >> 3;
    const int w = k * (total >> 3) + (k < (total & 7) ? k : (total & 7)) + j;
    bx = w / gn; by = w - bx * gn;
}

template <typename T, int BM, int BN, int NS, bool PRE = true>
__device__ __forceinline__ void nt_body(const NtParams& p, const int bx, const int by, const int wg_linear, const int ks = 0) {
#ifdef NT_STAMP
    unsigned long long stamp[16];
#endif
    NT_T(0);
    static_assert(BM == BN, "one feeder wave per operand: equal request counts");
    constexpr int WTM = BM / 2, WTN = BN / 2, MI = WTM / 16, NJ = WTN / 16;
    constexpr int IPT = BM / 8;                        // LDS-DMA requests per K tile per feeder wave (8 rows each)
    constexpr int STAGE = (BM + BN) * 128;
    static_assert((NS - 2) * IPT <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int m0 = bx * BM, n0 = by * BN;
    const RowDiv rd = make_rowdiv(m0, p.e.Le);
    // split-K: this workgroup owns K tiles [kt0, kt0 + KT) and the destination slab `ks`
    const int KT_all = (int)((p.Kb + 127) / 128);
    const int kt0 = p.ksplits > 1 ? ks * p.kt_per_split : 0;
    const int KT = p.ksplits > 1 ? (KT_all - kt0 < p.kt_per_split ? (KT_all - kt0 > 0 ? KT_all - kt0 : 0) : p.kt_per_split) : KT_all;
    EpiDev ev = p.e;                                   // (fields are wave-uniform scalars)
    if (p.ksplits > 1) ev.dst = (char*)ev.dst + (int64_t)ks * p.dst_split_stride * (ev.dst_dtype == C1V_F32 ? 4 : 2);

    if (wave >= 4) {
        // ---- feeder: lane (prow, ppos) of request i fills position ppos of tile row 8 i + prow, which must hold logical
        // chunk ppos ^ ((row>>1)&7) of that row (same 128-byte line, so coalescing is unchanged)
        const int prow = lane >> 3, ppos = lane & 7;
        const char* src[IPT];
        if (wave == 4) {
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int row = 8 * i + prow;
                int b, u;
                rowdiv(rd, m0 + row < ev.M ? row : ev.M - 1 - m0, b, u);
                src[i] = p.A + (int64_t)b * p.a_bstride_b + (int64_t)(p.a_off + u) * p.lda_b + ((ppos ^ ((row >> 1) & 7)) << 4);
            }
        } else {
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int row = 8 * i + prow;
                int n = n0 + row;
                if (n >= ev.N) n = ev.N - 1;
                src[i] = p.W + (int64_t)n * p.ldw_b + ((ppos ^ ((row >> 1) & 7)) << 4);
            }
        }
        const int dst0 = wave == 4 ? 0 : BM * 128;
        // byte offset of the next K tile in both operands' rows (tiles are issued strictly in order): NtParams::k_taps
        int tap_i = 0;
        int64_t ko = (int64_t)kt0 * 128;
        const int64_t tap_b = (int64_t)p.k_strips * 128;
        auto issue = [&](int kt) {
            char* d = smem + (kt % NS) * STAGE + dst0;
#if defined(NT_ABLATE) && (NT_ABLATE & 1)
            // diagnostic (wrong results): the A feeder only fetches tap 0 of every strip -- an upper bound for what an
            // activation-window kernel (one A fetch for all taps) could gain
            if (!(wave == 4 && p.k_taps > 1 && tap_i != 0))
#endif
#if defined(NT_ABLATE) && (NT_ABLATE & 2)
            if (kt < 0)                                    // diagnostic: no operand requests at all (compute side alone)
#endif
#pragma unroll
            for (int i = 0; i < IPT; ++i) glds16(src[i] + ko, d + i * 1024);
            if (p.k_taps > 1) {
                if (++tap_i == p.k_taps) { tap_i = 0; ko += 128 - (int64_t)(p.k_taps - 1) * tap_b; } else ko += tap_b;
            } else ko += 128;
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < KT) issue(s);
        for (int kt = 0; kt < KT; ++kt) {
            // tiles requested so far: min(KT, kt + NS - 1); all but the NS-2 youngest must have landed
            if (kt + NS - 2 < KT) wait_vmcnt<(NS - 2) * IPT>(); else wait_vmcnt<0>();
            block_barrier();                           // tile kt is in LDS; stage (kt-1) % NS is free again
            if (kt + NS - 1 < KT) issue(kt + NS - 1);
        }
        block_barrier();                               // the two barriers of the epilogue below
        block_barrier();
        return;
    }

    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // bias of this thread's epilogue channel group (see epilogue_rows): requested now, consumed after the K loop
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    const int fastw = ev.fastw;                       // wave-uniform: 4 / 2 = vectorised epilogue, 0 = element-wise
    EpiState es;
    {
        const int nbias = n0 + 4 * (tid % (BN / 4));
        if (ks > 0) {
            // split-K: the bias belongs to slab 0
        } else if (ev.bias && ev.bias_row_w > 0) {           // rows padded to reshape_w channels: bias index = t * bias_row_w + c
            const int tb = nbias / ev.reshape_w, cb = nbias - tb * ev.reshape_w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int t2 = tb, c2 = cb + k;
                if (c2 >= ev.reshape_w) { c2 -= ev.reshape_w; ++t2; }
                bias4[k] = (nbias + k < ev.N && c2 < ev.bias_row_w) ? ev.bias[t2 * ev.bias_row_w + c2] : 0.0f;
            }
        } else if (ev.bias) {
            if (fastw == 4) {
                if (nbias + 3 < ev.N) { const float4 b = *(const float4*)(ev.bias + nbias); bias4[0] = b.x; bias4[1] = b.y; bias4[2] = b.z; bias4[3] = b.w; }
                else {                                       // partial last group (rows padded beyond N): stay inside the bias vector
#pragma unroll
                    for (int k = 0; k < 4; ++k) bias4[k] = nbias + k < ev.N ? ev.bias[nbias + k] : 0.0f;
                }
            } else if (fastw == 2) {
                if (nbias < ev.N) { const float2 b = *(const float2*)(ev.bias + nbias); bias4[0] = b.x; bias4[1] = b.y; }
                if (nbias + 2 < ev.N) { const float2 b = *(const float2*)(ev.bias + nbias + 2); bias4[2] = b.x; bias4[3] = b.y; }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) bias4[k] = nbias + k < ev.N ? ev.bias[nbias + k] : 0.0f;
            }
        }
    }
    // epilogue inputs of the first chunk of rows: requested while the feeders fill the ring
    if constexpr (PRE) {
        if (fastw == 4) { C1V_EPI_DISPATCH(4, (epi_load<BM, BN, MODE_, VW_>(ev, rd, es, m0, n0, tid, 0))) }
        else if (fastw == 2) { C1V_EPI_DISPATCH(2, (epi_load<BM, BN, MODE_, VW_>(ev, rd, es, m0, n0, tid, 0))) }
    }
    NT_T(1);

    for (int kt = 0; kt < KT; ++kt) {
        block_barrier();
        const char* sa = smem + (kt % NS) * STAGE;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#if defined(NT_ABLATE) && (NT_ABLATE & 4)
            if (kt >= 0) continue;                         // diagnostic: no fragment reads, no MFMAs (feed side alone)
#endif
            uint4 af[MI], bf[NJ];
            const int ch = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const uint4*)(sa + lds_off(wm * WTM + i * 16 + (lane & 15), ch));
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = *(const uint4*)(sb + lds_off(wn * WTN + j * 16 + (lane & 15), ch));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) mma16<T>(acc[i][j], af[i], bf[j]);
        }
    }

    NT_T(2);
    // Epilogue through LDS.  acc[i][j][r] is C[m0 + wm*WTM + i*16 + (lane>>4)*4 + r][n0 + wn*WTN + j*16 + (lane&15)]; the
    // tile is parked in the (now dead) operand ring as fp32 and then walked by a ROLLED loop in which consecutive threads
    // own consecutive 4-channel groups of a row.  Two reasons: (1) code size -- the per-element epilogue unrolled over
    // the accumulator registers was ~40 KB of straight-line code executed once per wave, and its instruction-cache
    // misses cost more than the K loop at this model's shapes (measured with s_memtime stamps: 10-19 k cycles);
    // (2) stores and mask loads become whole 128-byte row segments instead of 32-byte scraps.
    constexpr int LDC = BN + 4;
    float* ct = (float*)smem;
    __syncthreads();                                   // every wave is done reading the last operand stage
    NT_T(4);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ct[(wm * WTM + i * 16 + (lane >> 4) * 4 + r) * LDC + wn * WTN + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
    NT_T(5);
    if constexpr (!PRE) {
        if (fastw == 4) { C1V_EPI_DISPATCH(4, (epi_seq_rows<BM, BN, MODE_, VW_>(ev, rd, ct, LDC, m0, n0, tid, bias4))) }
        else if (fastw == 2) { C1V_EPI_DISPATCH(2, (epi_seq_rows<BM, BN, MODE_, VW_>(ev, rd, ct, LDC, m0, n0, tid, bias4))) }
    } else if (fastw == 4) { C1V_EPI_DISPATCH(4, (epi_fast_rows<BM, BN, MODE_, VW_>(ev, rd, es, ct, LDC, m0, n0, tid, bias4))) }
    else if (fastw == 2) { C1V_EPI_DISPATCH(2, (epi_fast_rows<BM, BN, MODE_, VW_>(ev, rd, es, ct, LDC, m0, n0, tid, bias4))) }
    if (fastw != 0) {
    } else {
        switch (ev.row_mode) {                        // wave-uniform; one specialised epilogue each
            case C1V_ROW_STORE:   epilogue_rows<BM, BN, C1V_ROW_STORE>(ev, ct, LDC, m0, n0, tid, bias4); break;
            case C1V_ROW_POOL2:   epilogue_rows<BM, BN, C1V_ROW_POOL2>(ev, ct, LDC, m0, n0, tid, bias4); break;
            case C1V_ROW_UP2:     epilogue_rows<BM, BN, C1V_ROW_UP2>(ev, ct, LDC, m0, n0, tid, bias4); break;
            case C1V_ROW_PAIRSUM: epilogue_rows<BM, BN, C1V_ROW_PAIRSUM>(ev, ct, LDC, m0, n0, tid, bias4); break;
            default:              epilogue_rows<BM, BN, C1V_ROW_UNPOOL>(ev, ct, LDC, m0, n0, tid, bias4); break;
        }
    }
#ifdef NT_STAMP
    NT_T(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NT_T(3);
    const int wg = wg_linear;
    if (tid == 0 && wg < 8192) for (int i = 0; i < 8; ++i) nt_dbg[wg * 16 + i] = stamp[i];
#endif
}

// ------------------------------------------------------------------------------------------------------
// NT kernel of the LARGE convolutions (round 3; `ntw_body`, NtParams::wide): a 256 x 128 tile over the FLAT rows of the
// activation buffer, the activation window of a 128-byte channel strip fetched ONCE for all taps.
//
// Why: at C4's shapes the 128 x 128 kernel above is bound by what a CU takes in from L2, not by its MFMAs -- every K tile
// (one tap of one strip) brings 16 KiB of activations AND 16 KiB of weights for 512 MFMA cycles (64 B/clk wanted, ~30
// delivered), and the three taps of a strip fetch the same activation rows three times, one row further down each time.
// Here the GEMM rows are the buffer rows themselves (NtParams::e.Le = the sample pitch: rows between the samples compute
// garbage that the epilogue's `u < nrows` tests drop -- a few per cent at the layers this variant is taken for), so the
// activation rows of a tile are ONE contiguous window: BM + k - 1 rows, loaded once per strip into a ring of its own, and
// tap j reads its fragments j rows further down.  Per strip a CU then takes in 33 KiB of activations + 3 x 16 KiB of weights
// for 3 x 1024 MFMA cycles: 27 B/clk at the full MFMA rate instead of 64.
// Waves 0-3 compute 64 x 128 each (8 fragment reads per 32 MFMAs instead of 8 per 16), waves 4 / 5 feed the window ring /
// the weight ring.  One barrier per K tile; barrier t also promises that tile t+1 has LANDED, so the first fragments of the
// next tile are fetched before its barrier and no LDS latency is exposed behind one.  Fragment reads run three groups of
// four MFMAs ahead of their use (sched_barrier(0) pins the order: with one wave per SIMD nothing else hides LDS latency).
// Results are bit-identical to nt_body's (same products, same order of the K tiles).
// ------------------------------------------------------------------------------------------------------
// Two shapes of it (NtwCfg):
//   NtwBig  256 x 128 tile, waves 4 x 1 (64 x 128 each), weight ring of 5 with look-ahead (barrier t promises tile t+1), one
//           workgroup per CU.  Opt-in (C1V_NTW=big): its epilogues fall together on all CUs (see c1v_api.hip).
//   NtwTwo  128 x 128 tile, waves 2 x 2 (64 x 64 each), two-stage weight ring, <= 168 registers and 66 KiB of LDS: TWO
//           workgroups per CU that hide each other's epilogue and ring latency, like nt_body's large variant, but with a
//           third less to take in per K tile.
//   NtwRow  128 x 320 tile, waves 2 x 2 (64 x 160 each), three-stage weight ring, one workgroup per CU: the WHOLE output row of
//           a convolution with <= 320 channels (the output convolution: 302 word-vector components) lies in one workgroup,
//           whose epilogue (C1V_ROW_NORM) then also l2-normalises it -- the loss's first step without a pass of its own over
//           the fp32 rows.  The tile is parked in LDS in two halves of 64 rows (128 x 324 floats do not fit).
template <int BM_, int BN_, int WGM_, int WGN_, int NSW_, bool LOOK_, int PARK_ = BM_>
struct NtwCfg {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, NSW = NSW_, NSA = 2;
    static constexpr bool LOOK = LOOK_;
    static constexpr int MI = BM / WGM / 16, NJ = BN / WGN / 16;
    static constexpr int AROWS = BM + 8;                   // window rows: the tile's rows + up to 8 rows of halo (k <= 9)
    static constexpr int A_STAGE = AROWS * 128, W_STAGE = BN * 128;
    static constexpr int RING = NSA * A_STAGE + NSW * W_STAGE;
    static constexpr int LDC = BN + 4;
    static constexpr int PARK = PARK_;
    static constexpr int TAB = BN_ > 256 ? 2 * BM_ * 4 : 0;       // C1V_ROW_NORM: row sums behind the operand ring
    static constexpr int LDS = (RING > PARK_ * LDC * 4 ? RING : PARK_ * LDC * 4) + TAB;
    static constexpr int IPA = AROWS / 8, IPW = BN / 8;
    static_assert(WGM * WGN == 4 && MI <= NJ && NJ % 2 == 0 && AROWS % 8 == 0, "wave grid");
    static_assert(IPA <= 63 && (NSW - 2) * IPW <= 63 && LDS <= 160 * 1024, "vmcnt is a 6-bit counter; LDS of a CU");
};
using NtwBig = NtwCfg<256, 128, 4, 1, 5, true>;
using NtwTwo = NtwCfg<128, 128, 2, 2, 2, false>;
using NtwRow = NtwCfg<128, 320, 2, 2, 3, false, 64>;
using NtwRowS = NtwCfg<64, 320, 2, 2, 3, false, 64>;      // small launches: twice the workgroups, each stores half as much

template <typename T, int MODE, typename Cfg>
__device__ __forceinline__ void ntw_body(const NtParams& p, const int bx, const int by) {
#ifdef NT_STAMP
    unsigned long long stamp[16];
#endif
    NT_T(0);
    constexpr int BM = Cfg::BM, BN = Cfg::BN, MI = Cfg::MI, NJ = Cfg::NJ, NSW = Cfg::NSW;
    constexpr int IPA = Cfg::IPA, IPW = Cfg::IPW;
    constexpr bool LOOK = Cfg::LOOK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;
    char* const sW = smem + Cfg::NSA * Cfg::A_STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = bx * BM, n0 = by * BN;
    const RowDiv rd = make_rowdiv(m0, p.e.Le);
    const int taps = p.k_taps, strips = p.k_strips;
    const int KT = taps * strips;
    EpiDev ev = p.e;

    if (wave >= 4) {
        const int prow = lane >> 3, ppos = lane & 7;
        if (wave == 4) {
            // ---- activation windows: buffer rows a_off + m0 + [0, AROWS), clamped into the allocation (rows past the
            // last sample only feed GEMM rows >= M, which the epilogue drops)
            const int64_t lim = (int64_t)ev.M + C1V_SLACK_ROWS - 1;
            const char* src[IPA];
#pragma unroll
            for (int i = 0; i < IPA; ++i) {
                const int row = 8 * i + prow;
                int64_t r = (int64_t)p.a_off + m0 + row;
                if (r > lim) r = lim;
                src[i] = p.A + r * p.lda_b + ((ppos ^ ((row >> 1) & 7)) << 4);
            }
            auto issue = [&](int s) {
                char* d = sA + (s & 1) * Cfg::A_STAGE;
                const int64_t ko = (int64_t)s * 128;
#pragma unroll
                for (int i = 0; i < IPA; ++i) glds16(src[i] + ko, d + i * 1024);
            };
            issue(0);
            if (strips > 1) issue(1);
            int tap = 0, s = 0;                            // tile t = (strip s, tap)
            for (int t = 0; t < KT; ++t) {
                // barrier t promises the window of tile t (LOOK: of tile t+1): strip 0 at t = 0, afterwards a new strip
                // whenever that tile is a tap 0
                if (t == 0) { if (strips > 1) wait_vmcnt<IPA>(); else wait_vmcnt<0>(); }
                else if (LOOK ? tap == taps - 1 : tap == 0) wait_vmcnt<0>();
                block_barrier();
                // everybody is done with tile t-1: at the first tile of strip s >= 1 the stage of strip s-1 is free for s+1
                if (tap == 0 && s >= 1 && s + 1 < strips) issue(s + 1);
                if (++tap == taps) { tap = 0; ++s; }
            }
        } else {
            // ---- weight tiles: rows n0 + [0, BN), K tile t = (strip t / taps, tap t % taps) at byte offset tap*k_strips*128 + strip*128
            const char* src[IPW];
#pragma unroll
            for (int i = 0; i < IPW; ++i) {
                const int row = 8 * i + prow;
                int n = n0 + row;
                if (n >= ev.N) n = ev.N - 1;
                src[i] = p.W + (int64_t)n * p.ldw_b + ((ppos ^ ((row >> 1) & 7)) << 4);
            }
            int tap_i = 0;
            int64_t ko = 0;
            const int64_t tap_b = (int64_t)strips * 128;
            auto issue = [&](int kt) {
                char* d = sW + (kt % NSW) * Cfg::W_STAGE;
#pragma unroll
                for (int i = 0; i < IPW; ++i) glds16(src[i] + ko, d + i * 1024);
                if (++tap_i == taps) { tap_i = 0; ko += 128 - (int64_t)(taps - 1) * tap_b; } else ko += tap_b;
            };
#pragma unroll
            for (int t = 0; t < NSW - 1; ++t)
                if (t < KT) issue(t);
            for (int t = 0; t < KT; ++t) {
                // issued so far: tiles < min(KT, t + NSW - 1); tiles <= `need` must have landed
                const int issued = t + NSW - 1 < KT ? t + NSW - 1 : KT;
                const int need = (LOOK && t + 1 < KT) ? t + 1 : t;
                const int out = issued - 1 - need;         // tiles that may still be in flight
                if (out >= 2) wait_vmcnt<(NSW >= 5 ? 2 : 0) * IPW>();
                else if (out == 1) wait_vmcnt<(NSW >= 4 || (NSW >= 3 && !LOOK) ? 1 : 0) * IPW>();
                else wait_vmcnt<0>();
                block_barrier();                           // everybody is done with tile t-1: its stage takes tile t+NSW-1
                if (t + NSW - 1 < KT) issue(t + NSW - 1);
            }
        }
        block_barrier();                                   // the two barriers of the epilogue below
        block_barrier();
        if constexpr (MODE == C1V_ROW_NORM && Cfg::PARK < BM) { block_barrier(); block_barrier(); }      // (two per parked part there)
        return;
    }

    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // bias of this thread's epilogue channel group (see epilogue_rows): requested now, consumed after the K loop
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    // ROW_NORM: the bias of this lane's accumulator columns (wn * BN/WGN + 16 j + (lane & 15); 0 past N)
    float biasj[NJ];
    constexpr int NP = BM / Cfg::PARK;                     // the tile is parked in NP parts of PARK rows
    int nrm_pos_[NP], nrm_b_[NP], nrm_u_[NP];
    if constexpr (MODE == C1V_ROW_NORM) {
        const int wn_ = (wave - (wave / Cfg::WGN) * Cfg::WGN) * (BN / Cfg::WGN);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = wn_ + 16 * j + (lane & 15);
            biasj[j] = (ev.bias && c < ev.N) ? ev.bias[c] : 0.0f;
        }
        // lane l < 16 looks after rows wave + 4 l of every part: (sample, step) and the row's compact position (-2: not
        // a row of the convolution, -1: a row without a target), requested now and used after the K loop
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const int r = h * Cfg::PARK + wave + 4 * (lane & 15);
            nrm_pos_[h] = -2;
            rowdiv(rd, r, nrm_b_[h], nrm_u_[h]);
            if (m0 + r < ev.M && nrm_u_[h] < ev.nrows && nrm_u_[h] < ev.dst_len) {
                const int ridx = nrm_b_[h] * ev.nrows + nrm_u_[h];
                const int q = ev.nrm_pos ? ev.nrm_pos[ridx] : ridx;
                nrm_pos_[h] = q < 0 ? -1 : q;
            }
        }
    } else {
        const int nbias = n0 + 4 * (tid % (BN / 4));
        if (ev.bias && ev.bias_row_w > 0) {
            const int tb = nbias / ev.reshape_w, cb = nbias - tb * ev.reshape_w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int t2 = tb, c2 = cb + k;
                if (c2 >= ev.reshape_w) { c2 -= ev.reshape_w; ++t2; }
                bias4[k] = (nbias + k < ev.N && c2 < ev.bias_row_w) ? ev.bias[t2 * ev.bias_row_w + c2] : 0.0f;
            }
        } else if (ev.bias) {
#pragma unroll
            for (int k = 0; k < 4; ++k) bias4[k] = nbias + k < ev.N ? ev.bias[nbias + k] : 0.0f;
        }
    }

    // fragment addresses: A fragment i of tap j = window row (wave's first row) + 16 i + (lane & 15) + j, chunk kk*4 + (lane >> 4);
    // the swizzle depends on the row through (row >> 1) & 7 only, which steps of 16 rows leave alone, and the second half of
    // the K tile (kk = 1) is the first one's address with bit 6 flipped
    const int fr = lane & 15, fc = lane >> 4;
    const int wm = wave / Cfg::WGN, wn = wave - wm * Cfg::WGN;
    // (byte offsets from `smem`; every stage offset is a multiple of 128, so bit 6 belongs to the chunk index)
    const int a_wave = wm * (BM / Cfg::WGM) * 128;
    const int b_lane = Cfg::NSA * Cfg::A_STAGE + wn * (BN / Cfg::WGN) * 128 + lds_off(fr, fc);
    auto a_addr = [&](int s, int tap) -> int { return (s & 1) * Cfg::A_STAGE + a_wave + lds_off(fr + tap, fc); };
    auto b_addr = [&](int t) -> int { return (t % NSW) * Cfg::W_STAGE + b_lane; };
    auto ld = [&](int base, int frag, int kk) -> uint4 { return *(const uint4*)(smem + ((base ^ (kk * 64)) + frag * 16 * 128)); };

    uint4 af[2][MI], bf[4];
    NT_T(1);
    if constexpr (LOOK) {
        block_barrier();                                   // barrier 0: tiles 0 and 1 are in LDS
        const int a0 = a_addr(0, 0), b0 = b_addr(0);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = ld(a0, i, 0);
#pragma unroll
        for (int j = 0; j < 3; ++j) bf[j] = ld(b0, j, 0);
    }
    NT_T(7);
    int tap = 0, s = 0;
    for (int t = 0; t < KT; ++t) {
        const int a_cur = a_addr(s, tap), b_cur = b_addr(t);
        int tapn = tap + 1, sn = s;
        if (tapn == taps) { tapn = 0; ++sn; }
        const bool more = LOOK && t + 1 < KT;              // wave-uniform
        const int a_nxt = a_addr(sn, tapn), b_nxt = b_addr(t + 1);
        if constexpr (!LOOK) {
            block_barrier();                               // barrier t: tile t is in LDS
#pragma unroll
            for (int i = 0; i < MI; ++i) af[0][i] = ld(a_cur, i, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[j] = ld(b_cur, j, 0);
        }
#pragma unroll
        for (int g = 0; g < 2 * NJ; ++g) {
            const int kk = g / NJ, j = g % NJ;
            // reads that run ahead: the weight fragment of group g+3 and, one per group, the activation fragments of the
            // next half tile (groups 0..MI-1: kk = 1 of this tile; groups NJ..NJ+MI-1, LOOK only: kk = 0 of the next tile,
            // which barrier t has promised)
            const int gn = g + 3;
            if (gn < 2 * NJ) bf[gn & 3] = ld(b_cur, gn % NJ, gn / NJ);
            else if (more) bf[gn & 3] = ld(b_nxt, gn - 2 * NJ, 0);
            if (g < MI) af[1][g] = ld(a_cur, g, 1);
            else if (g >= NJ && g < NJ + MI && more) af[0][g - NJ] = ld(a_nxt, g - NJ, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) mma16<T>(acc[i][j], af[kk][i], bf[g & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        tap = tapn; s = sn;
        if (more) block_barrier();                         // barrier t+1
    }

    // ---- epilogue through LDS (see nt_body): the tile parked as fp32, walked by rows
    NT_T(2);
    float* ct = (float*)smem;
    if constexpr (MODE == C1V_ROW_NORM) {
        // 1. v = acc + bias (0 in the columns past N); sum of squares of every row from the registers: a lane squares its NJ
        //    columns of a row, four DPP steps add the 16 lanes that share the row, the two column halves meet in LDS (behind
        //    the operand ring: written before the barrier that ends the K loop)
        // 2. the tile is parked in LDS in parts of PARK rows; wave w of the four walks rows w, w + 4, ... of the part, a lane
        //    holds columns 4 lane .. 4 lane + 3 (and 256 + 4 lane ..): 16-byte stores of the fp32 row, 8-byte (bf16) stores of
        //    the normalised row.  The walk is bound by what a CU can store (measured 15-25 B/clk: 119 KB per 64 rows), not
        //    by its instructions: dealing 16-byte chunks across rows (29 full store instructions instead of 64) was slower.
        constexpr int HR = Cfg::PARK, RPW = HR / 4;
        static_assert(Cfg::WGM == 2 && Cfg::WGN == 2 && (NP == 1 || NP == 2) && BN > 256 && BN <= 512 && RPW == 16 && Cfg::TAB >= 2 * BM * 4, "row walk");
        T* const yhat = (T*)ev.nrm_yhat;
        float* const ydst = (float*)ev.dst;
        float* const ssum = (float*)(smem + Cfg::LDS - Cfg::TAB);      // [2][BM]
        {
            const int c0 = wn * (BN / Cfg::WGN) + (lane & 15);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bool in = c0 + 16 * j < ev.N;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = in ? acc[i][j][r] + biasj[j] : 0.0f;
                        acc[i][j][r] = v;
                        sq[r] += v * v;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = sq[r];
                    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
                    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
                    sq[r] = v;
                }
                if ((lane & 15) == 0)
                    *(float4*)(ssum + wn * BM + wm * (BM / 2) + i * 16 + (lane >> 4) * 4) = make_float4(sq[0], sq[1], sq[2], sq[3]);
            }
        }
        block_barrier();                                   // operand stages are done with; the row sums are complete
        NT_T(4);
        const bool tail = lane < (BN - 256) / 4;
#pragma unroll 1
        for (int part = 0; part < NP; ++part) {
            if (part) { NT_T(5); block_barrier(); }        // the rows of the part before are done with
            if (NP == 1 || wm == part) {
                // (NP == 1: waves wm = 0 / 1 hold rows 0.. / BM/2.. of the one part)
                const int r0 = NP == 1 ? wm * (BM / 2) : 0;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            ct[(r0 + i * 16 + (lane >> 4) * 4 + r) * Cfg::LDC + wn * (BN / Cfg::WGN) + j * 16 + (lane & 15)] = acc[i][j][r];
            }
            const int my_b = part ? nrm_b_[NP - 1] : nrm_b_[0], my_u = part ? nrm_u_[NP - 1] : nrm_u_[0];
            const int my_pos = part ? nrm_pos_[NP - 1] : nrm_pos_[0];
            float my_rr;
            {
                const int r = part * HR + wave + 4 * (lane & (RPW - 1));
                my_rr = 1.0f / sqrtf(fmaxf(ssum[r] + ssum[BM + r], 1e-12f));
                if (lane < RPW && my_pos >= 0) ev.nrm_rnorm[my_b * ev.nrows + my_u] = my_rr;
            }
            const int64_t my_y = (int64_t)my_b * ev.dst_bstride + (int64_t)(ev.dst_off + my_u) * ev.dst_cp;
            block_barrier();
            // rows in groups of four: the LDS reads of a group are issued together, then its stores
#pragma unroll
            for (int l0 = 0; l0 < RPW; l0 += 4) {
                float4 va[4], vb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* cr = ct + (wave + 4 * (l0 + k)) * Cfg::LDC + 4 * lane;
                    va[k] = *(const float4*)cr;
                    vb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (tail) vb[k] = *(const float4*)(cr + 256);
                }
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int l = l0 + k4;
                    const int pos = __builtin_amdgcn_readlane(my_pos, l);
                    const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_rr), l)) * ev.nrm_qscale;
                    const int64_t yo = ((int64_t)__builtin_amdgcn_readlane((int)(my_y >> 32), l) << 32) |
                                       (uint32_t)__builtin_amdgcn_readlane((int)my_y, l);
                    const float4 a4 = va[k4], b4 = vb[k4];
                    // (pos == -2: not a row of the convolution -- between samples / past the end)
                    if (pos == -1 && ev.nrm_dy != nullptr) {
                        // a row without a target: its loss gradient is zero, written here so that the finish step can run
                        // over the live rows only
                        const int b_ = __builtin_amdgcn_readlane(my_b, l), u_ = __builtin_amdgcn_readlane(my_u, l);
                        if (16 * lane < ev.nrm_dy_row_b)
                            *(uint4*)(ev.nrm_dy + (int64_t)b_ * ev.nrm_dy_bstride_b + (int64_t)(ev.nrm_dy_off + u_) * ev.nrm_dy_row_b + 16 * lane) =
                                make_uint4(0u, 0u, 0u, 0u);
                    }
                    if (pos >= 0 || (pos == -1 && !ev.nrm_live_only)) {
                        float* yr = ydst + yo + 4 * lane;
                        if (4 * lane < ev.dst_cp) *(float4*)yr = a4;
                        if (tail && 256 + 4 * lane < ev.dst_cp) *(float4*)(yr + 256) = b4;
                    }
                    if (pos >= 0) {
                        const int64_t ho = (int64_t)pos * ev.nrm_kp + 4 * lane;
                        if constexpr (sizeof(T) == 2) {
                            if (4 * lane < ev.nrm_kp)
                                *(uint2*)((bf16_raw*)yhat + ho) = make_uint2(pack_bf16x2(a4.x * sc, a4.y * sc), pack_bf16x2(a4.z * sc, a4.w * sc));
                            if (tail && 256 + 4 * lane < ev.nrm_kp)
                                *(uint2*)((bf16_raw*)yhat + ho + 256) = make_uint2(pack_bf16x2(b4.x * sc, b4.y * sc), pack_bf16x2(b4.z * sc, b4.w * sc));
                        } else {
                            if (4 * lane < ev.nrm_kp) *(float4*)((float*)yhat + ho) = make_float4(a4.x * sc, a4.y * sc, a4.z * sc, a4.w * sc);
                            if (tail && 256 + 4 * lane < ev.nrm_kp)
                                *(float4*)((float*)yhat + ho + 256) = make_float4(b4.x * sc, b4.y * sc, b4.z * sc, b4.w * sc);
                        }
                    }
                }
            }
        }
    } else {
    block_barrier();                                       // every wave is done reading the last operand stages
    NT_T(4);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ct[(wm * (BM / Cfg::WGM) + i * 16 + (lane >> 4) * 4 + r) * Cfg::LDC + wn * (BN / Cfg::WGN) + j * 16 + (lane & 15)] = acc[i][j][r];
    block_barrier();
    NT_T(5);
    // (one row mode per instantiation, 4-wide vector epilogue only -- the host takes this variant for such launches only: with
    // every mode in one kernel the epilogue was 26 k instructions of straight-line code and ~3 k SGPR spill moves)
    epi_seq_rows<BM, BN, MODE, 4>(ev, rd, ct, Cfg::LDC, m0, n0, tid, bias4);
    }
#ifdef NT_STAMP
    NT_T(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NT_T(3);
    if (tid == 0 && blockIdx.x < 8192) for (int i = 0; i < 8; ++i) nt_dbg[blockIdx.x * 16 + i] = stamp[i];
#endif
}

// ------------------------------------------------------------------------------------------------------
// TN kernel (wgrad): 128x128 output tile, K loop over rows r; operands are read "down the columns":
//   bf16: ds_read_b64_tr_b16 (hardware transpose), fragment element 4w+q <-> row 16w + 4g + q of a 32-row step
//   fp32: ds_read_b32 straight from the row-major tile (v_mfma_f32_16x16x4_f32 wants one k per lane group)
// Row pitch = 128*esize + pad (bf16 +32 B, fp32 +64 B) keeps both read patterns conflict-free.  A tile image is
// BR rows = 18 KiB, filled by 18 one-KiB LDS-DMA requests of that operand's feeder wave (waves 4 / 5, see nt_body).
// ------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------
// wgrad epilogue: one 128 x 128 accumulator tile -> out[ii * si + jj * sj] through LDS.  Straight from the MFMA layout
// every store instruction of the old epilogue wrote 4 x 16 scattered values (64 different 4-byte addresses when the
// contiguous index of the slab was the one the accumulator holds in its REGISTERS -- the (k,1,F,C) layout of the transposed
// convolutions); at C4, with 196 KB of slab per workgroup, that store tail was longer than the K loop.  The tile is parked
// in LDS with the memory-contiguous index fastest and written out 256 consecutive bytes per wave instruction.
// All 256 compute threads call it; `ct` = TN_CT_LD * 128 floats of LDS that nobody reads any more.
// ------------------------------------------------------------------------------------------------------
constexpr int TN_CT_LD = 132;
__device__ __forceinline__ void tn_park_tile(const TnParams& p, float* ct, const f32x4_t (&acc)[4][4], int wi, int wj, int lane) {
    if (p.sj == 1) {                                     // (wave-uniform) j contiguous in memory: ct[i][j]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ct[(wi * 64 + i * 16 + (lane >> 4) * 4 + r) * TN_CT_LD + wj * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    } else {                                             // i contiguous: ct[j][i], the four registers of a fragment as one word
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(float4*)(ct + (wj * 64 + j * 16 + (lane & 15)) * TN_CT_LD + wi * 64 + i * 16 + (lane >> 4) * 4) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
}
__device__ __forceinline__ void tn_write_tile(const TnParams& p, const float* ct, float* out, int i0, int j0, int tid) {
    const bool jfast = p.sj == 1;
    const int n_major = jfast ? p.I - i0 : p.J - j0, n_minor = jfast ? p.J - j0 : p.I - i0;     // valid extents of this tile
    const int64_t s_major = jfast ? p.si : p.sj, s_minor = jfast ? p.sj : p.si;
    float* o = out + (int64_t)(jfast ? i0 : j0) * s_major + (int64_t)(jfast ? j0 : i0) * s_minor;
    const int minor = tid & 127, m0 = tid >> 7;
    if (minor >= n_minor) return;
#pragma unroll 8
    for (int e = 0; e < 64; ++e) {
        const int major = 2 * e + m0;
        if (major < n_major) o[(int64_t)major * s_major + (int64_t)minor * s_minor] = ct[major * TN_CT_LD + minor];
    }
}

template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_raw> { static constexpr int BR = 64, PITCH = 128 * 2 + 32; };
template <> struct TnCfg<float>    { static constexpr int BR = 32, PITCH = 128 * 4 + 64; };
constexpr int TN_TILE = 18 * 1024;
constexpr int TN_NS = 2;

// NS: depth of the operand ring.  2 (72 KiB, two workgroups per CU hide each other's request latency) everywhere but in the
// batched launch of small problems, where a CU holds one workgroup and a K tile then takes one request round trip
// (~2 k cycles for 512 cycles of MFMAs): 4 stages keep three tiles in flight.
template <typename T, int NS = TN_NS>
__device__ __forceinline__ void tn_body(const TnParams& p, const int bx, const int by, const int bz) {
    using Cfg = TnCfg<T>;
    constexpr int BR = Cfg::BR, PITCH = Cfg::PITCH;
    constexpr int ES = sizeof(T);
    constexpr int CPR = PITCH / 16, DCH = 128 * ES / 16;
    constexpr int STAGE = 2 * TN_TILE;
    constexpr int NPIECE = BR * PITCH / 1024;         // LDS-DMA requests (1 KiB each) per tile image
    static_assert(BR * PITCH == 18 * 1024 && NPIECE * 64 == BR * CPR, "tile image is 18 KiB of whole requests");
    static_assert((NS - 1) * NPIECE <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = bx * 128, j0 = by * 128;
    const int tap = bz / p.splits, split = bz - tap * p.splits;
    const int64_t rb = p.r_begin + (int64_t)split * p.rows_per_split;
    int64_t re = rb + p.rows_per_split;
    if (re > p.r_end) re = p.r_end;
    const int64_t nrows = re > rb ? re - rb : 0;
    const int KT = (int)((nrows + BR - 1) / BR);

    const char* abase = p.A + (rb + p.a_shift0 + tap) * p.lda_b + (int64_t)i0 * ES;
    const char* dbase = p.D + rb * p.ldd_b + (int64_t)j0 * ES;

    if (wave >= 4) {
        // ---- feeder (wave 4: A image, wave 5: D image): per-lane sources as byte offsets from the first row of a tile;
        // lanes that land in the row padding read the row's first chunk
        const bool feed_a = wave == 4;
        const int64_t ld_b = feed_a ? p.lda_b : p.ldd_b;
        const int col0_b = (feed_a ? i0 : j0) * ES, cols_b = (feed_a ? p.Ia : p.Ja) * ES;
        int64_t off[NPIECE];
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int Lc = i * 64 + lane;
            const int row = Lc / CPR;
            int pos = Lc - row * CPR;
            if (pos >= DCH || col0_b + pos * 16 >= cols_b) pos = 0;
            off[i] = (int64_t)row * ld_b + pos * 16;
        }
        const char* base = feed_a ? abase : dbase;
        const int dst0 = feed_a ? 0 : TN_TILE;
        auto issue = [&](int kt) {
            char* d = smem + (kt % NS) * STAGE + dst0;
            const char* t = base + (int64_t)kt * BR * ld_b;
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) glds16(t + off[i], d + i * 1024);
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < KT) issue(s);
        for (int kt = 0; kt < KT; ++kt) {
            // issued so far: tiles < min(KT, kt + NS - 1); tile kt must have landed, the later ones may be in flight
            const int fly = (kt + NS - 1 < KT ? kt + NS - 1 : KT) - (kt + 1);
            if (NS >= 4 && fly >= 2) wait_vmcnt<(NS >= 4 ? 2 : 0) * NPIECE>();
            else if (NS >= 3 && fly >= 1) wait_vmcnt<(NS >= 3 ? 1 : 0) * NPIECE>();
            else wait_vmcnt<0>();
            block_barrier();
            if (kt + NS - 1 < KT) issue(kt + NS - 1);
        }
        block_barrier(); block_barrier(); block_barrier();     // the barriers of the epilogue below
        return;
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t accb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accb[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.bias_out != nullptr && bx == 0 && tap == 0 && wi == 0;     // wave-uniform

    for (int kt = 0; kt < KT; ++kt) {
        block_barrier();
        const char* sa = smem + (kt % NS) * STAGE;
        const char* sd = sa + TN_TILE;
        if constexpr (sizeof(T) == 2) {
            const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
#pragma unroll
            for (int ks = 0; ks < BR / 32; ++ks) {
                uint4 af[4], df[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const char* a0 = sa + (ks * 32 + 4 * g + q) * PITCH + (wi * 64 + i * 16 + 4 * pq) * 2;
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(C1V_LDS_PTR(s16x4_t, a0));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(C1V_LDS_PTR(s16x4_t, a0 + 16 * PITCH));
                    af[i] = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y,
                                       __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const char* d0 = sd + (ks * 32 + 4 * g + q) * PITCH + (wj * 64 + j * 16 + 4 * pq) * 2;
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(C1V_LDS_PTR(s16x4_t, d0));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(C1V_LDS_PTR(s16x4_t, d0 + 16 * PITCH));
                    df[j] = make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y,
                                       __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16<bf16_raw>(acc[i][j], af[i], df[j]);
                if (do_bias) {
                    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16<bf16_raw>(accb[j], ones, df[j]);
                }
            }
        } else {
            for (int ks = 0; ks < BR / 4; ++ks) {
                float af[4], df[4];
                const int row = ks * 4 + (lane >> 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const float*)(sa + row * PITCH + (wi * 64 + i * 16 + (lane & 15)) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) df[j] = *(const float*)(sd + row * PITCH + (wj * 64 + j * 16 + (lane & 15)) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], df[j], acc[i][j], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, df[j], accb[j], 0, 0, 0);
                }
            }
        }
    }

    const int tap_dst = p.flip ? (p.taps - 1 - tap) : tap;
    float* ct = (float*)smem;
    block_barrier();                                       // every wave is done reading the last operand stage
    tn_park_tile(p, ct, acc, wi, wj, lane);
    block_barrier();
    tn_write_tile(p, ct, p.out + (int64_t)split * p.slab_stride + (int64_t)tap_dst * p.tap_stride, i0, j0, tid);
    block_barrier();
    if (do_bias && (lane >> 4) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jj = j0 + wj * 64 + j * 16 + (lane & 15);
            if (jj < p.J) p.bias_out[(int64_t)split * p.bias_slab_stride + jj] = accb[j][0];
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// TN kernel with the taps of a convolution FUSED into one workgroup (round 3).  The per-tap form above re-reads both
// operands once per tap (and D once per 128 input channels on top): at C4's shapes the wgrad launches were bound by the
// bytes a CU can take in from L2 (72 B/clk wanted at the MFMA rate, ~30 B/clk delivered), not by the MFMAs.  Here one
// workgroup owns the same 128 x 128 (input channel x output channel) block of TG <= 3 consecutive taps: the A image
// carries BR + TG - 1 rows, tap t reads it t rows further down, and the D fragments of a 32-row step feed the MFMAs of
// every tap -- a third of the bytes in per MFMA and 16 instead of 24 fragment reads per 48 MFMAs.  The price is
// TG x 64 accumulator registers per wave, i.e. one workgroup per CU, which a three-stage ring makes up for.
// grid.z = ceil(taps / TG) * splits.
// ------------------------------------------------------------------------------------------------------
constexpr int TNF_A_IMG = 20 * 1024;                // BR + 2 rows of PITCH bytes, rounded up to whole 1 KiB requests
constexpr int TNF_STAGE = TNF_A_IMG + TN_TILE;
constexpr int TNF_NS = 3;
constexpr int TNF_LDS = TNF_NS * TNF_STAGE;

template <typename T, int TG>
__device__ __forceinline__ void tn_taps_body(const TnParams& p, const int bx, const int by, const int bz) {
    using Cfg = TnCfg<T>;
    constexpr int BR = Cfg::BR, PITCH = Cfg::PITCH;
    constexpr int ES = sizeof(T);
    constexpr int CPR = PITCH / 16, DCH = 128 * ES / 16;
    constexpr int NPA = TNF_A_IMG / 1024, NPD = BR * PITCH / 1024;
    constexpr int AROWS = BR + TG - 1;
    static_assert(TG >= 1 && TG <= 3 && AROWS * PITCH <= TNF_A_IMG && NPD * 1024 == BR * PITCH, "tile images");
    static_assert((TNF_NS - 2) * NPA <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = bx * 128, j0 = by * 128;
    const int tgi = bz / p.splits, split = bz - tgi * p.splits;
    const int tap0 = tgi * TG;
    // (taps is a multiple of TG: host check)
    const int64_t rb = p.r_begin + (int64_t)split * p.rows_per_split;
    int64_t re = rb + p.rows_per_split;
    if (re > p.r_end) re = p.r_end;
    const int64_t nrows = re > rb ? re - rb : 0;
    const int KT = (int)((nrows + BR - 1) / BR);

    const char* abase = p.A + (rb + p.a_shift0 + tap0) * p.lda_b + (int64_t)i0 * ES;
    const char* dbase = p.D + rb * p.ldd_b + (int64_t)j0 * ES;

    if (wave >= 4) {
        // ---- feeder (wave 4: A image, wave 5: D image); lanes that land in the row padding or behind the image's last
        // row read the tile's first chunk
        const bool feed_a = wave == 4;
        const int64_t ld_b = feed_a ? p.lda_b : p.ldd_b;
        const int col0_b = (feed_a ? i0 : j0) * ES, cols_b = (feed_a ? p.Ia : p.Ja) * ES;
        const int rows_img = feed_a ? AROWS : BR;
        int64_t off[NPA];
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int Lc = i * 64 + lane;
            int row = Lc / CPR;
            int pos = Lc - row * CPR;
            if (pos >= DCH || col0_b + pos * 16 >= cols_b) pos = 0;
            if (row >= rows_img) { row = 0; pos = 0; }
            off[i] = (int64_t)row * ld_b + pos * 16;
        }
        const char* base = feed_a ? abase : dbase;
        const int dst0 = feed_a ? 0 : TNF_A_IMG;
        auto issue = [&](int kt) {
            char* d = smem + (kt % TNF_NS) * TNF_STAGE + dst0;
            const char* t = base + (int64_t)kt * BR * ld_b;
#if defined(TN_ABLATE) && (TN_ABLATE & 1)
            if (kt < 0)                                    // diagnostic: no operand requests at all (compute side alone)
#endif
            if (feed_a) {
#pragma unroll
                for (int i = 0; i < NPA; ++i) glds16(t + off[i], d + i * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < NPD; ++i) glds16(t + off[i], d + i * 1024);
            }
        };
#pragma unroll
        for (int s = 0; s < TNF_NS - 1; ++s)
            if (s < KT) issue(s);
        for (int kt = 0; kt < KT; ++kt) {
            // tiles requested so far: min(KT, kt + NS - 1); all but the NS-2 youngest must have landed
            if (kt + TNF_NS - 2 < KT) { if (feed_a) wait_vmcnt<(TNF_NS - 2) * NPA>(); else wait_vmcnt<(TNF_NS - 2) * NPD>(); }
            else wait_vmcnt<0>();
            block_barrier();
            if (kt + TNF_NS - 1 < KT) issue(kt + TNF_NS - 1);
        }
#pragma unroll
        for (int b = 0; b < 1 + 2 * TG; ++b) block_barrier();     // the barriers of the epilogue below
        return;
    }

    f32x4_t acc[TG][4][4];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x4_t accb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accb[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.bias_out != nullptr && bx == 0 && tgi == 0 && wi == 0;     // wave-uniform

    for (int kt = 0; kt < KT; ++kt) {
        block_barrier();
        const char* sa = smem + (kt % TNF_NS) * TNF_STAGE;
        const char* sd = sa + TNF_A_IMG;
#if defined(TN_ABLATE) && (TN_ABLATE & 2)
        if (kt >= 0) continue;                             // diagnostic: no fragment reads, no MFMAs (feed side alone)
#endif
        if constexpr (sizeof(T) == 2) {
            // One wave per SIMD: nothing but this wave's own instruction stream hides the LDS latency, so the fragment reads run
            // one GROUP (two A fragments = 8 MFMAs = 128 cycles) ahead of the MFMAs that use them; sched_barrier(0) pins
            // "reads of group g+1, then MFMAs of group g" (left alone, the compiler put every read in front of its first use:
            // one LDS round trip per four MFMAs, 3.6 x the MFMA time at C4).  The D fragments of the next 32-row step are
            // requested right behind the last MFMAs of this one and arrive while those drain.
            const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
            const char* arow = sa + (4 * g + q) * PITCH + (wi * 64 + 4 * pq) * 2;
            const char* drow = sd + (4 * g + q) * PITCH + (wj * 64 + 4 * pq) * 2;
            constexpr int KS = BR / 32, NG = 2 * TG;
            auto ldf = [&](const char* p0) -> uint4 {
                s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(C1V_LDS_PTR(s16x4_t, p0));
                s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(C1V_LDS_PTR(s16x4_t, p0 + 16 * PITCH));
                return make_uint4(__builtin_bit_cast(uint2, lo).x, __builtin_bit_cast(uint2, lo).y,
                                  __builtin_bit_cast(uint2, hi).x, __builtin_bit_cast(uint2, hi).y);
            };
            uint4 df[4], af[2][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) df[j] = ldf(drow + j * 32);
#pragma unroll
            for (int e = 0; e < 2; ++e) af[0][e] = ldf(arow + e * 32);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const int t = gi >> 1, i0 = (gi & 1) * 2;
                    if (gi + 1 < NG) {
                        const int tn = (gi + 1) >> 1, in0 = ((gi + 1) & 1) * 2;
#pragma unroll
                        for (int e = 0; e < 2; ++e) af[(gi + 1) & 1][e] = ldf(arow + (ks * 32 + tn) * PITCH + (in0 + e) * 32);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int j = 0; j < 4; ++j) mma16<bf16_raw>(acc[t][i0 + e][j], af[gi & 1][e], df[j]);
                    if (gi == 0 && do_bias) {
                        const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
#pragma unroll
                        for (int j = 0; j < 4; ++j) mma16<bf16_raw>(accb[j], ones, df[j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (gi + 1 == NG && ks + 1 < KS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) df[j] = ldf(drow + (ks + 1) * 32 * PITCH + j * 32);
#pragma unroll
                        for (int e = 0; e < 2; ++e) af[0][e] = ldf(arow + (ks + 1) * 32 * PITCH + e * 32);
                    }
                }
            }
        } else {
            const char* arow = sa + (lane >> 4) * PITCH + (wi * 64 + (lane & 15)) * 4;
            const char* drow = sd + (lane >> 4) * PITCH + (wj * 64 + (lane & 15)) * 4;
#pragma unroll 2
            for (int ks = 0; ks < BR / 4; ++ks) {
                float df[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) df[j] = *(const float*)(drow + ks * 4 * PITCH + j * 64);
#pragma unroll
                for (int t = 0; t < TG; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float af = *(const float*)(arow + (ks * 4 + t) * PITCH + i * 64);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, df[j], acc[t][i][j], 0, 0, 0);
                    }
                }
                if (do_bias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, df[j], accb[j], 0, 0, 0);
                }
            }
        }
    }

    float* ct = (float*)smem;
    block_barrier();                                       // every wave is done reading the last operand stage
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const int tap = tap0 + t;
        const int tap_dst = p.flip ? (p.taps - 1 - tap) : tap;
        tn_park_tile(p, ct, acc[t], wi, wj, lane);
        block_barrier();
        tn_write_tile(p, ct, p.out + (int64_t)split * p.slab_stride + (int64_t)tap_dst * p.tap_stride, i0, j0, tid);
        block_barrier();
    }
    if (do_bias && (lane >> 4) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jj = j0 + wj * 64 + j * 16 + (lane & 15);
            if (jj < p.J) p.bias_out[(int64_t)split * p.bias_slab_stride + jj] = accb[j][0];
        }
    }
}

// LDS needs and tile choice, shared by the launchers of c1v_gemm.hip and c1v_gemm_pair.hip
constexpr int TN_LDS = TN_NS * 2 * TN_TILE;
static_assert(TN_LDS >= TN_CT_LD * 128 * 4 && TNF_LDS >= TN_CT_LD * 128 * 4, "the parked wgrad tile must fit the operand ring");
enum NtVariant { NT_128 = 0, NT_128L = 1, NT_64 = 2 };
// LDS of an NT workgroup: the operand ring, reused by the epilogue for the parked fp32 tile (BM x (BN + 4))
template <int BM, int BN, int NS> constexpr int nt_lds() {
    return NS * (BM + BN) * 128 > BM * (BN + 4) * 4 ? NS * (BM + BN) * 128 : BM * (BN + 4) * 4;
}
static inline int64_t nt_large_tiles() {
    // (diagnostic / test override: the tile count from which the large-GEMM variant is taken)
    // (default: measured per layer at C4, profiles/r02b_c4_gemm_variant_threshold.txt -- from one tile per CU on, two
    // co-resident workgroups with shallow rings beat one with a deep ring; round 1 started at 768)
    const char* env = getenv("C1V_NT_LARGE_TILES");
    return env ? atoll(env) : 256;
}
static inline int nt_large_ns() {
    // ring depth of the large-GEMM variant: 2 stages = 64 KiB = two workgroups per CU; 4 / 5 stages = one workgroup per CU
    // with 3 / 4 tiles in flight (diagnostic override)
    const char* env = getenv("C1V_NT_LARGE_NS");
    return env ? atoi(env) : 2;
}
static inline NtVariant nt_variant(const NtParams& p) {
    // Measured (s_memtime stamps, compute removed): a CU takes in ~30 B/clk whatever the depth of the ring, so the K loops
    // of these small GEMMs are bound by the bytes ONE workgroup needs; shallow rings that let two workgroups share a CU
    // (and a wgrad + dgrad pair share the chip) beat deep ones.
    // Large GEMMs (several rounds of 128 x 128 tiles on every CU; the C4 stack at batch 2048): NT_128L = the same tile with
    // a two-stage ring (64 KiB) and <= 168 registers, so that TWO workgroups share a CU and one's epilogue (as long as its
    // K loop at these K) runs under the other's K loop.  Split-K launches and the latent head's dgrad (M = batch: always
    // small) keep the small-GEMM variants.
    const int64_t tiles128 = c1v_cdiv(p.e.M, 128) * c1v_cdiv(p.e.N, 128);
    if (p.ksplits <= 1 && p.e.row_mode != C1V_ROW_LATENT_BWD && tiles128 >= nt_large_tiles()) return NT_128L;
    return tiles128 >= 384 ? NT_128 : NT_64;      // (NT_128: split-K / latent-head launches of that size, and threshold overrides)
}
template <typename K> static void set_lds(K kern, int bytes) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
