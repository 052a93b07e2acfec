// HBM-bound kernels of libc1dvae: operand packing, latent head, bias-gradient column sums, gradient finalize,
// Nadam + global-norm clip, EOS truncation, Philox RNG.  All are grid-stride, coalesced, and reduction orders are
// fixed (no float atomics) so that two runs of a step are bit-identical.
#include "c1v_common.h"
#include <stdlib.h>
#include "c1v_lossred.h"

// index of the descriptor that owns block `blk`: descriptors are sorted by first_block (the first one starts at 0).
// All lanes of a wave look at one descriptor each, so the search is one round of independent loads per 64 descriptors
// (a serial scan cost one dependent global-load latency per descriptor).
template <typename D>
__device__ __forceinline__ int find_desc(const D* __restrict__ descs, int n_desc, int blk) {
    const int lane = threadIdx.x & 63;
    int found = 0;
    for (int base = 0; base < n_desc; base += 64) {
        const int i = base + lane;
        const bool le = i < n_desc && descs[i].first_block <= blk;
        found += __popcll(__ballot(le));
    }
    return found - 1;
}

// ------------------------------------------------------------------------------------------------------
// input packing
// ------------------------------------------------------------------------------------------------------
// one thread per (row, group of 8 padded channels): 8 coalesced fp32 reads, one 16- / 32-byte store (zeros in the padding)
template <typename T>
__device__ __forceinline__ void store8(c1v_seq dst, int64_t idx, const float v[8]) {
    if constexpr (sizeof(T) == 2) {
        *(uint4*)((bf16_raw*)dst.data + idx) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    } else {
        float4* d = (float4*)((float*)dst.data + idx);
        d[0] = make_float4(v[0], v[1], v[2], v[3]); d[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// (blk, nblk): this block's index among the blocks of the job -- the jobs also run as block ranges of c1v_step_prologue
template <typename T>
__device__ __forceinline__ void pack_input_body(const float* __restrict__ src, int T_len, int C, const c1v_seq& dst, int blk, int nblk) {
    const int g8 = dst.cp >> 3;
    const int64_t total = (int64_t)dst.batch * T_len * g8;
    for (int64_t i = blk * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
        int c0, t;
        int64_t row, b;
        if (total < (1ll << 31)) {                     // 32-bit index arithmetic (64-bit divisions are several times slower)
            const uint32_t i32 = (uint32_t)i, r32 = i32 / (uint32_t)g8, b32 = r32 / (uint32_t)T_len;
            c0 = (int)(i32 - r32 * (uint32_t)g8) * 8; row = r32; t = (int)(r32 - b32 * (uint32_t)T_len); b = b32;
        } else {
            c0 = (int)(i % g8) * 8; row = i / g8; t = (int)(row % T_len); b = row / T_len;
        }
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {              // unconditional load (clamped), masked afterwards: no branch, no wait per element
            const float t2 = src[row * C + (c0 + k < C ? c0 + k : C - 1)];
            v[k] = c0 + k < C ? t2 : 0.0f;
        }
        store8<T>(dst, b * dst.bstride + (int64_t)(dst.off + t) * dst.cp + c0, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ src, int T_len, int C, c1v_seq dst) {
    pack_input_body<T>(src, T_len, C, dst, (int)blockIdx.x, (int)gridDim.x);
}

template <typename T>
__device__ __forceinline__ void gather_input_body(const float* __restrict__ table, int64_t ld, const int* __restrict__ ids,
                                                  int T_len, int C, const c1v_seq& dst, int blk, int nblk) {
    const int g8 = dst.cp >> 3;
    const int64_t total = (int64_t)dst.batch * T_len * g8;
    const bool vec = (ld & 3) == 0 && (((uintptr_t)table) & 15) == 0;
    for (int64_t i = blk * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)nblk * blockDim.x) {
        int c0, t;
        int64_t row, b;
        if (total < (1ll << 31)) {
            const uint32_t i32 = (uint32_t)i, r32 = i32 / (uint32_t)g8, b32 = r32 / (uint32_t)T_len;
            c0 = (int)(i32 - r32 * (uint32_t)g8) * 8; row = r32; t = (int)(r32 - b32 * (uint32_t)T_len); b = b32;
        } else {
            c0 = (int)(i % g8) * 8; row = i / g8; t = (int)(row % T_len); b = row / T_len;
        }
        const int id = ids[row];
        float v[8];
        if (vec && c0 + 8 <= ld) {                 // two 16-byte words of the table row (rows padded to whole words)
            const float4* p = (const float4*)(table + (int64_t)(id >= 0 ? id : 0) * ld + c0);
            const float4 a = p[0], b4 = p[1];
            const float t8[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (c0 + k < C && id >= 0) ? t8[k] : 0.0f;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t2 = table[(int64_t)(id >= 0 ? id : 0) * ld + (c0 + k < C ? c0 + k : C - 1)];
                v[k] = (c0 + k < C && id >= 0) ? t2 : 0.0f;
            }
        }
        store8<T>(dst, b * dst.bstride + (int64_t)(dst.off + t) * dst.cp + c0, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gather_input_kernel(const float* __restrict__ table, int64_t ld, const int* __restrict__ ids,
                                                           int T_len, int C, c1v_seq dst) {
    gather_input_body<T>(table, ld, ids, T_len, C, dst, (int)blockIdx.x, (int)gridDim.x);
}

static unsigned grid_for(int64_t n) {
    int64_t g = c1v_cdiv(n, 256);
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

static int check_seq(const c1v_seq& s, const char* what) {
    C1V_REQUIRE(s.data != nullptr, "%s: null buffer", what);
    C1V_REQUIRE(s.batch > 0 && s.cp > 0 && s.len > 0 && s.off >= 0 && s.c > 0 && s.c <= s.cp, "%s: bad descriptor", what);
    C1V_REQUIRE(s.bstride >= (int64_t)(s.off + s.len) * s.cp, "%s: sample stride too small", what);
    C1V_REQUIRE(s.dtype == C1V_F32 || s.dtype == C1V_BF16, "%s: bad dtype", what);
    return 0;
}

extern "C" int c1v_pack_input(const c1v_pack_input_args* a, void* stream) {
    C1V_REQUIRE(a && a->src, "pack_input: null argument");
    if (check_seq(a->dst, "pack_input dst")) return -1;
    C1V_REQUIRE(a->T > 0 && a->T <= a->dst.len && a->C > 0 && a->C <= a->dst.cp, "pack_input: shape mismatch (T=%d C=%d)", a->T, a->C);
    C1V_REQUIRE(a->dst.cp % 8 == 0 && a->dst.bstride % 8 == 0 && ((uintptr_t)a->dst.data % 32) == 0, "pack_input: dst rows must be 8-element groups, 32-byte aligned");
    const int64_t total = (int64_t)a->dst.batch * a->T * (a->dst.cp / 8);
    if (a->dst.dtype == C1V_F32) hipLaunchKernelGGL(pack_input_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a->src, a->T, a->C, a->dst);
    else hipLaunchKernelGGL(pack_input_kernel<bf16_raw>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a->src, a->T, a->C, a->dst);
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_gather_input(const c1v_gather_input_args* a, void* stream) {
    C1V_REQUIRE(a && a->table && a->ids, "gather_input: null argument");
    if (check_seq(a->dst, "gather_input dst")) return -1;
    C1V_REQUIRE(a->T > 0 && a->T <= a->dst.len && a->C > 0 && a->C <= a->dst.cp && a->ld >= a->C, "gather_input: shape mismatch");
    C1V_REQUIRE(a->dst.cp % 8 == 0 && a->dst.bstride % 8 == 0 && ((uintptr_t)a->dst.data % 32) == 0, "gather_input: dst rows must be 8-element groups, 32-byte aligned");
    const int64_t total = (int64_t)a->dst.batch * a->T * (a->dst.cp / 8);
    if (a->dst.dtype == C1V_F32) hipLaunchKernelGGL(gather_input_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a->table, a->ld, a->ids, a->T, a->C, a->dst);
    else hipLaunchKernelGGL(gather_input_kernel<bf16_raw>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a->table, a->ld, a->ids, a->T, a->C, a->dst);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// weight operand copies (descriptor table on the device; blockIdx.y = descriptor)
// ------------------------------------------------------------------------------------------------------
constexpr int PW_TS = C1V_PACK_TILE;                 // tile edge of the weight packer (include/c1dvae.h)
__device__ __forceinline__ void pack_weights_body(const c1v_pack_desc* __restrict__ descs, int n_desc, int blk) {
    // One block = one 64(c) x 64(n) tile of one tap of one descriptor, transposed through LDS so that BOTH the fp32 read
    // (along whichever of n / c is contiguous in the Keras layout: 256-byte row segments) and the operand write (along
    // c) are whole row segments.  Blocks are dealt to descriptors by `first_block`; within a descriptor the tile index
    // enumerates (tap, c tile, n tile).  Loads are unconditional (clamped index, value masked): a load under a bounds
    // test compiles to a branch with its own wait.
    __shared__ float tile[PW_TS][PW_TS + 1];
    const int di = find_desc(descs, n_desc, blk);
    const c1v_pack_desc d = descs[di];
    const int tc = (d.c_count + PW_TS - 1) / PW_TS, tn = (d.n_count + PW_TS - 1) / PW_TS;
    int bi = blk - d.first_block;
    const int tn_i = bi % tn; bi /= tn;
    const int tc_i = bi % tc;
    const int tap = bi / tc;
    if (tap >= d.taps) return;
    const int tx = threadIdx.x & (PW_TS - 1), ty = threadIdx.x / PW_TS;          // 64 x 4
    constexpr int RPT = PW_TS / (256 / PW_TS);                                   // rows per thread
    const int c0 = tc_i * PW_TS, n0 = tn_i * PW_TS;
    const float* src = d.src + (int64_t)tap * d.s_tap;
    float v[RPT];
    if (d.s_n <= d.s_c) {
        // n is the fast axis of the source: read rows of n, tile[c][n]
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int c = c0 + ty + (256 / PW_TS) * k, n = n0 + tx;
            v[k] = src[(int64_t)(c < d.c_count ? c : d.c_count - 1) * d.s_c + (int64_t)(n < d.n_count ? n : d.n_count - 1) * d.s_n];
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) tile[ty + (256 / PW_TS) * k][tx] = v[k];
    } else {
        // c is the fast axis of the source: read rows of c
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int n = n0 + ty + (256 / PW_TS) * k, c = c0 + tx;
            v[k] = src[(int64_t)(c < d.c_count ? c : d.c_count - 1) * d.s_c + (int64_t)(n < d.n_count ? n : d.n_count - 1) * d.s_n];
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) tile[tx][ty + (256 / PW_TS) * k] = v[k];
    }
    __syncthreads();
    const int tp = d.flip ? d.taps - 1 - tap : tap;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int n = n0 + ty + (256 / PW_TS) * k, c = c0 + tx;
        if (c < d.c_count && n < d.n_count) store_any(d.dst, d.dtype, (int64_t)n * d.ldk + (int64_t)tp * d.cp + c, tile[tx][ty + (256 / PW_TS) * k]);
    }
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const c1v_pack_desc* __restrict__ descs, int n_desc) {
    pack_weights_body(descs, n_desc, (int)blockIdx.x);
}

extern "C" int c1v_pack_weights(const c1v_pack_desc* descs, int32_t n_desc, int32_t total_blocks, void* stream) {
    C1V_REQUIRE(descs && n_desc > 0 && total_blocks > 0, "pack_weights: bad argument");
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, descs, n_desc);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// latent head: sampling (:1188-1191), KL (:1202), decoder-input dropout (:1254) and their backward
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void latent_fwd_kernel(c1v_latent_args a) {
    // one wave per sample; the columns go in rounds of 4 x 64 with every load of a round requested before the first value
    // is used (clamped index, result masked): a plain `for (d = lane; d < D; d += 64)` loop paid one memory round trip per
    // 64 columns, which was most of this small kernel's time
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const float* m = a.mlv + (int64_t)b * 2 * a.D;
    const float* lv = m + a.D;
    const float* ep = a.eps + (int64_t)b * a.D;
    const bool kd = a.keep_d != nullptr;
    // (the mask bytes are loaded unconditionally -- from eps when there is no mask, and ignored: a load under a condition
    // compiles to a branch that waits for it at once, one exposed memory round trip per round)
    const uint8_t* kdp = kd ? a.keep_d + (int64_t)b * a.D : (const uint8_t*)ep;
    float kl = 0.f;
    for (int d0 = 0; d0 < a.D; d0 += 256) {
        float mv[4], l[4], e[4];
        uint8_t k[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = d0 + lane + 64 * r, dc = d < a.D ? d : a.D - 1;
            mv[r] = m[dc]; l[r] = lv[dc]; e[r] = ep[dc];
            k[r] = kdp[dc];
        }
        if (a.n_parts > 1) {
            // split-K heads GEMM: mlv arrives as partial slabs; sum them in slab order and write the sum back to slab 0
            for (int s2 = 1; s2 < a.n_parts; s2 += 4) {          // four slabs (32 requests) per round
                float pm[4][4], pl[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t so = (int64_t)(s2 + i < a.n_parts ? s2 + i : a.n_parts - 1) * a.part_stride;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = d0 + lane + 64 * r, dc = d < a.D ? d : a.D - 1;
                        pm[i][r] = m[so + dc]; pl[i][r] = lv[so + dc];
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (s2 + i < a.n_parts) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { mv[r] += pm[i][r]; l[r] += pl[i][r]; }
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = d0 + lane + 64 * r;
                if (d < a.D) { ((float*)m)[d] = mv[r]; ((float*)lv)[d] = l[r]; }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = d0 + lane + 64 * r;
            if (d < a.D) {
                const float z = mv[r] + expf(a.std_scale * l[r]) * e[r];
                kl += 1.0f + l[r] - mv[r] * mv[r] - expf(l[r]);
                if (a.z_out) a.z_out[(int64_t)b * a.D + d] = z;
                float zd = z;
                if (kd) zd = k[r] ? z / a.keep_prob : 0.0f;
                store_any(a.zd_out, a.dtype, (int64_t)b * a.ldz + d, zd);
            }
        }
    }
    kl = wave_sum(kl);
    if (lane == 0 && a.kl_out) a.kl_out[b] = kl / (float)a.D;
}
__global__ __launch_bounds__(256) void latent_bwd_kernel(c1v_latent_args a) {
    const int64_t total = (int64_t)a.B * a.D;
    const float kc = 1.0f / ((float)a.D * (float)a.B_global);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % a.D);
        const int64_t b = i / a.D;
        const float mv = a.mlv[b * 2 * a.D + d], l = a.mlv[b * 2 * a.D + a.D + d];
        float dz = a.dzd[i];
        if (a.keep_d) dz = a.keep_d[i] ? dz / a.keep_prob : 0.0f;
        const float dm = dz + 2.0f * mv * kc;
        const float dl = dz * a.eps[i] * a.std_scale * expf(a.std_scale * l) + (expf(l) - 1.0f) * kc;
        store_any(a.dmlv, a.dtype, b * a.ldd + d, dm);
        store_any(a.dmlv, a.dtype, b * a.ldd + a.D + d, dl);
    }
}

extern "C" int c1v_latent_head_fwd(const c1v_latent_args* a, void* stream) {
    C1V_REQUIRE(a && a->mlv && a->eps && a->zd_out, "latent_head_fwd: null argument");
    C1V_REQUIRE(a->B > 0 && a->D > 0 && a->ldz >= a->D && a->keep_prob > 0.f, "latent_head_fwd: bad shape");
    hipLaunchKernelGGL(latent_fwd_kernel, dim3((unsigned)c1v_cdiv(a->B, 4)), dim3(256), 0, (hipStream_t)stream, *a);
    C1V_LAUNCH_CHECK();
    return 0;
}
extern "C" int c1v_latent_head_bwd(const c1v_latent_args* a, void* stream) {
    C1V_REQUIRE(a && a->mlv && a->eps && a->dzd && a->dmlv, "latent_head_bwd: null argument");
    C1V_REQUIRE(a->B > 0 && a->D > 0 && a->ldd >= 2 * a->D && a->B_global > 0 && a->keep_prob > 0.f, "latent_head_bwd: bad shape");
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(grid_for((int64_t)a->B * a->D)), dim3(256), 0, (hipStream_t)stream, *a);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// c1v_dec_head_fwd: latent head -> decoder dense -> first transposed convolution in ONE launch (small batches).
// As launches of their own the three are 6.4 + 6.1 + 5.2 us at C2 and almost all of that is latency: launch, argument
// fetch, one memory round trip in, one store drain out, three times over, for 0.1 GFLOP.  Here a workgroup owns DH_G samples
// from the partial sums of the heads GEMM to the up-sampled output of the transposed convolution; what the stages hand to
// each other stays in LDS (the buffers the backward pass reads are written on the side), and both weight operands are
// requested as MFMA fragments before the first stage starts.  Same MFMA instruction, same order of the K steps and the
// same epilogue arithmetic as the GEMM kernels: results are bit-identical to the three launches.
// ------------------------------------------------------------------------------------------------------
constexpr int DH_G = 4;          // samples per workgroup
constexpr int DH_W = 16;         // waves: four per SIMD -- every stage is short chains of dependent memory / LDS operations, and
                                 // with one wave per SIMD (first version: 41 k cycles per workgroup, 15 k of them spent ISSUING
                                 // 144 loads per wave through the 63-deep counter) nothing hides them
constexpr int DH_NBW = 2;        // dense: 16-column blocks per wave            (n_out <= 16 * 2 * 16 = 512)
constexpr int DH_KSD = 8;        // dense: 32-element K steps                   (K <= 256)
constexpr int DH_RBW = 4;        // transposed convolution: 16-row blocks per wave, two wave groups  (DH_G * n_rows <= 128)
constexpr int DH_KSC = 4;        //   32-element K steps                        (k * cp <= 128)
                                 //   one 16-column block per wave, eight blocks (n_out <= 128)
struct DecHeadDev {
    c1v_latent_args lat;
    const bf16_raw* wd; int64_t ldwd; int Kd, Nd, reshape_w, bias_row_w, relu_d; const float* bias_d;
    bf16_raw* xd; int64_t xd_bstride; int xd_off, xd_len, xd_cp, pitch;
    const bf16_raw* wc; int64_t ldwc; int k, n_rows, F, relu_c; const float* bias_c;
    bf16_raw* xf; int64_t xf_bstride; int xf_off, xf_len, xf_cp;
    int lds_img, lds_park, lds_kl, lds_total;  // byte offsets: input image of the convolution, its parked output, KL partial sums
};
// (LDS hand-offs inside the workgroup: wait for this wave's LDS writes only, then the barrier.  __syncthreads() would also
// wait for every global load and store in flight -- the weight fragments requested ahead, the side stores of the buffers)
__device__ __forceinline__ void dh_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#ifdef DH_STAMP
// diagnostic build only: s_memtime stamps of wave 0 of every workgroup
__device__ unsigned long long dh_dbg[16 * 1024];
#define DH_T(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i]) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int c1v_debug_read_dh(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dh_dbg), sizeof(unsigned long long) * n);
}
#else
#define DH_T(i) do {} while (0)
#endif
__global__ __launch_bounds__(64 * DH_W) void dec_head_fwd_kernel(const DecHeadDev p) {
#ifdef DH_STAMP
    unsigned long long stamp[16];
#endif
    DH_T(0);
    c1v_touch_kernarg<6>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int b0 = blockIdx.x * DH_G;
    bf16_raw* zd_img = (bf16_raw*)smem;                        // [16][Kd]   dense operand (rows >= DH_G stay zero)
    bf16_raw* x_img = (bf16_raw*)(smem + p.lds_img);           // [DH_G][pitch][cp]  = the samples' rows of the buffer `xd`
    bf16_raw* park = (bf16_raw*)(smem + p.lds_park);           // [DH_G * n_rows][F]
    float* klp = (float*)(smem + p.lds_kl);                    // [DH_G][4][64]
    // ---- latent head: wave w = columns lane + 64 (w / 4) of sample b0 + w % 4 (c1v_latent_head_fwd's arithmetic for D <= 256 and
    // <= 8 partial slabs, same order of the additions).  Its loads are requested first: they are the critical path, the weight
    // fragments behind them arrive while the latent values are formed.  No load sits under a condition (a conditional load
    // compiles to a branch that waits for it at once): indices are clamped, values masked.
    const c1v_latent_args& a = p.lat;
    const int es = wave & 3, q = wave >> 2;
    const int b = b0 + es;
    const bool have_b = b < a.B;
    const int bcl = have_b ? b : a.B - 1;
    const int d = lane + 64 * q, dc = d < a.D ? d : a.D - 1;
    const bool have_d = have_b && d < a.D;
    const float* m = a.mlv + (int64_t)bcl * 2 * a.D;
    const float* lv = m + a.D;
    const float* ep = a.eps + (int64_t)bcl * a.D;
    const bool kd = a.keep_d != nullptr;
    const uint8_t* kdp = kd ? a.keep_d + (int64_t)bcl * a.D : (const uint8_t*)ep;     // (no mask: bytes of eps, ignored)
    float mv = m[dc], l = lv[dc];
    const float e1 = ep[dc];
    const uint8_t k1 = kdp[dc];
    float pm[7], pl[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int64_t so = (int64_t)(1 + i < a.n_parts ? 1 + i : (a.n_parts > 0 ? a.n_parts - 1 : 0)) * a.part_stride;
        pm[i] = m[so + dc]; pl[i] = lv[so + dc];
    }
    // the epilogues' bias values
    float bias_dv[DH_NBW], bias_cv;
    int t_d[DH_NBW], c_d[DH_NBW];
    {
        const float* bdp = p.bias_d ? p.bias_d : a.eps;
#pragma unroll
        for (int i = 0; i < DH_NBW; ++i) {
            const int n = (wave + DH_W * i) * 16 + r, nc = n < p.Nd ? n : p.Nd - 1;
            const int t = nc / p.reshape_w, c = nc - t * p.reshape_w;
            t_d[i] = t; c_d[i] = c;
            const int bi = p.bias_row_w > 0 ? t * p.bias_row_w + (c < p.bias_row_w ? c : p.bias_row_w - 1) : nc;
            bias_dv[i] = bdp[p.bias_d ? bi : 0];             // (no bias: one readable float, ignored)
            if (!p.bias_d || (p.bias_row_w > 0 && c >= p.bias_row_w)) bias_dv[i] = 0.f;
        }
        const float* bcp = p.bias_c ? p.bias_c : a.eps;
        const int f = (wave & 7) * 16 + r;
        bias_cv = bcp[p.bias_c ? (f < p.F ? f : p.F - 1) : 0];
        if (!p.bias_c) bias_cv = 0.f;
    }
    // both weight operands as MFMA fragments (lane (r, g): operand row = column block * 16 + r, 8 K values from 8 g), in the
    // order the K steps consume them.  Dense: wave w owns column blocks w and w + 16; transposed convolution: column block
    // w % 8 and the row blocks w / 8, w / 8 + 2, ...
    const int ksd = p.Kd >> 5, ksc = (p.k * p.xd_cp) >> 5;
    uint4 bd[DH_NBW][DH_KSD], bc[DH_KSC];
#pragma unroll
    for (int s2 = 0; s2 < DH_KSD; ++s2)
#pragma unroll
        for (int i = 0; i < DH_NBW; ++i) {
            const int n = (wave + DH_W * i) * 16 + r;
            const bf16_raw* src = p.wd + (int64_t)(n < p.Nd ? n : p.Nd - 1) * p.ldwd + 8 * g;
            bd[i][s2] = *(const uint4*)(src + 32 * (s2 < ksd ? s2 : ksd - 1));      // (clamped; surplus blocks / steps are never used)
        }
    {
        const int f = (wave & 7) * 16 + r;
        const bf16_raw* src = p.wc + (int64_t)(f < p.F ? f : p.F - 1) * p.ldwc + 8 * g;
#pragma unroll
        for (int s2 = 0; s2 < DH_KSC; ++s2) bc[s2] = *(const uint4*)(src + 32 * (s2 < ksc ? s2 : ksc - 1));
    }
    DH_T(1);
    for (int i = tid * 16; i < p.lds_park; i += 64 * DH_W * 16) *(uint4*)(smem + i) = make_uint4(0u, 0u, 0u, 0u);
    dh_barrier();
    DH_T(2);
    if (a.n_parts > 1) {
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (1 + i < a.n_parts) { mv += pm[i]; l += pl[i]; }
        if (have_d) { ((float*)m)[d] = mv; ((float*)lv)[d] = l; }
    }
    {
        float klq = 0.f;
        if (have_d) {
            const float z = mv + expf(a.std_scale * l) * e1;
            klq = 1.0f + l - mv * mv - expf(l);
            if (a.z_out) a.z_out[(int64_t)b * a.D + d] = z;
            float zd = z;
            if (kd) zd = k1 ? z / a.keep_prob : 0.0f;
            const bf16_raw h = f32_to_bf16(zd);
            ((bf16_raw*)a.zd_out)[(int64_t)b * a.ldz + d] = h;
            zd_img[es * p.Kd + d] = h;
        }
        klp[(es * 4 + q) * 64 + lane] = klq;
    }
    DH_T(3);
    dh_barrier();
    DH_T(4);
    if (q == 0 && have_b && a.kl_out) {
        // per lane in column-quarter order, then across the lanes: the order of the one-wave-per-sample kernel
        float kl = 0.f;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
            if (lane + 64 * qq < a.D) kl += klp[(es * 4 + qq) * 64 + lane];
        kl = wave_sum(kl);
        if (lane == 0) a.kl_out[b] = kl / (float)a.D;
    }
    // ---- dense: out(b, n) = sum_k zd(b, k) * wd(n, k); rows of the MFMA tile = samples (DH_G of 16 used)
    f32x4_t accd[DH_NBW];
#pragma unroll
    for (int i = 0; i < DH_NBW; ++i) accd[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (wave * 16 < p.Nd) {
#pragma unroll
        for (int s2 = 0; s2 < DH_KSD; ++s2) {
            if (s2 < ksd) {
                const uint4 af = *(const uint4*)(zd_img + r * p.Kd + 32 * s2 + 8 * g);
#pragma unroll
                for (int i = 0; i < DH_NBW; ++i)
                    accd[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bd[i][s2]), accd[i], 0, 0, 0);
            }
        }
    }
    DH_T(5);
    // epilogue (c1v_epilogue: bias, relu, (t, c) = divmod(n, reshape_w)) into the image; accd[i][e] = out(sample 4 g + e, n)
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < DH_NBW; ++i) {
            const int n = (wave + DH_W * i) * 16 + r;
            if (n < p.Nd && t_d[i] < p.xd_len) {
                const int o = (p.xd_off + t_d[i]) * p.xd_cp + c_d[i];
#pragma unroll
                for (int e = 0; e < DH_G; ++e) {
                    float v = accd[i][e] + bias_dv[i];
                    if (p.relu_d) v = fmaxf(v, 0.0f);
                    x_img[e * p.pitch * p.xd_cp + o] = f32_to_bf16(v);
                }
            }
        }
    }
    DH_T(6);
    dh_barrier();
    // the valid rows of the image -> the buffer (the backward pass reads it): whole rows, the channels behind reshape_w are zero
    {
        const int pr = p.xd_cp >> 3;                             // 16-byte pieces per row
        const int per = p.xd_len * pr, total = DH_G * per;
        for (int i = tid; i < total; i += 64 * DH_W) {
            const int e = i / per, q2 = i - e * per;
            const int o = (e * p.pitch + p.xd_off) * p.xd_cp + 8 * q2;       // (rows of a sample are contiguous: piece q2 of them)
            if (b0 + e < a.B)
                *(uint4*)(p.xd + (int64_t)(b0 + e) * p.xd_bstride + p.xd_off * p.xd_cp + 8 * q2) = *(const uint4*)(x_img + o);
        }
    }
    DH_T(7);
    // ---- transposed convolution = convolution over the zero-bordered image: GEMM row m = e * n_rows + u, K = taps x channels
    const int rows_c = DH_G * p.n_rows;
    const int cb = wave & 7, rb0 = wave >> 3;
    f32x4_t accc[DH_RBW];
#pragma unroll
    for (int jb = 0; jb < DH_RBW; ++jb) accc[jb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (cb * 16 < p.F) {
#pragma unroll
        for (int jb = 0; jb < DH_RBW; ++jb) {
            const int rb = rb0 + 2 * jb;
            if (rb * 16 < rows_c) {
                int mr = rb * 16 + r;
                if (mr >= rows_c) mr = rows_c - 1;
                const int e = mr / p.n_rows, u = mr - e * p.n_rows;
                const bf16_raw* arow = x_img + (e * p.pitch + u) * p.xd_cp + 8 * g;      // the window starts at buffer row u
#pragma unroll
                for (int s2 = 0; s2 < DH_KSC; ++s2) {
                    if (s2 < ksc) {
                        const uint4 af = *(const uint4*)(arow + 32 * s2);
                        accc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bc[s2]), accc[jb], 0, 0, 0);
                    }
                }
            }
        }
    }
    DH_T(8);
    {
        const int f = cb * 16 + r;
        if (f < p.F) {
#pragma unroll
            for (int jb = 0; jb < DH_RBW; ++jb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int mr = (rb0 + 2 * jb) * 16 + 4 * g + e;
                    if (mr < rows_c) {
                        float v = accc[jb][e] + bias_cv;
                        if (p.relu_c) v = fmaxf(v, 0.0f);
                        park[mr * p.F + f] = f32_to_bf16(v);
                    }
                }
        }
    }
    DH_T(9);
    dh_barrier();
    // C1V_ROW_UP2: output row t of a sample = convolution row t / 2 (rows behind 2 * n_rows are not written)
    {
        const int pr = p.F >> 3;
        const int per = p.xf_len * pr, total = DH_G * per;
        for (int i = tid; i < total; i += 64 * DH_W) {
            const int e = i / per, q2 = i - e * per;
            const int t = q2 / pr, pc = q2 - t * pr, u = t >> 1;
            if (b0 + e < a.B && u < p.n_rows)
                *(uint4*)(p.xf + (int64_t)(b0 + e) * p.xf_bstride + (int64_t)(p.xf_off + t) * p.xf_cp + 8 * pc) =
                    *(const uint4*)(park + (e * p.n_rows + u) * p.F + 8 * pc);
        }
    }
#ifdef DH_STAMP
    DH_T(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DH_T(11);
    if (tid == 0 && blockIdx.x < 1024) for (int i = 0; i < 12; ++i) dh_dbg[blockIdx.x * 16 + i] = stamp[i];
#endif
}

static int dec_head_fill(const c1v_dec_head_args* a, DecHeadDev& p) {
    C1V_REQUIRE(a && a->latent && a->dense && a->deconv, "dec_head_fwd: null argument");
    const c1v_latent_args& l = *a->latent;
    const c1v_dense_args& d = *a->dense;
    const c1v_conv1d_args& c = *a->deconv;
    C1V_REQUIRE(l.mlv && l.eps && l.zd_out && l.B > 0 && l.D > 0 && l.ldz >= l.D && l.keep_prob > 0.f, "dec_head_fwd: bad latent arguments");
    C1V_REQUIRE(l.dtype == C1V_BF16 && d.dtype == C1V_BF16 && c.x.dtype == C1V_BF16, "dec_head_fwd: bf16 operands only");
    C1V_REQUIRE(l.D <= 256 && l.n_parts <= 8 && (l.n_parts <= 1 || l.part_stride > 0), "dec_head_fwd: latent_dim <= 256 and at most 8 partial slabs");
    C1V_REQUIRE(d.a == l.zd_out && d.lda == l.ldz && d.rows == l.B && d.K == d.lda, "dec_head_fwd: the dense operand must be the latent head's zd rows");
    C1V_REQUIRE(d.K % 32 == 0 && d.K <= 32 * DH_KSD && d.ldw >= d.K && d.w_op && d.k_splits <= 1 && d.n_out > 0 && d.n_out <= 16 * DH_W * DH_NBW,
                "dec_head_fwd: dense shape (K %d, n_out %d) outside the fused kernel's limits", d.K, d.n_out);
    const c1v_epilogue& ed = d.epi;
    C1V_REQUIRE(ed.row_mode == C1V_ROW_STORE && ed.reshape_w > 0 && ed.act.data == nullptr && ed.keep == nullptr && ed.sel == nullptr &&
                ed.dst.dtype == C1V_BF16 && ed.bias_row_w >= 0 && ed.bias_row_w <= ed.reshape_w, "dec_head_fwd: dense epilogue must be a reshaping bf16 STORE");
    C1V_REQUIRE(ed.dst.data == c.x.data && ed.dst.bstride == c.x.bstride && ed.dst.cp == c.x.cp && ed.dst.off == c.x.off && ed.dst.batch == l.B && c.x.batch == l.B,
                "dec_head_fwd: the dense epilogue must write the transposed convolution's input buffer");
    C1V_REQUIRE(ed.reshape_w <= ed.dst.cp && d.n_out <= ed.dst.len * ed.reshape_w, "dec_head_fwd: dense rows do not fit the buffer");
    C1V_REQUIRE(c.w_op && c.k >= 1 && c.a_off == 0 && c.x.off == c.k - 1 && c.n_rows > 0 && c.x.cp % 32 == 0 && c.k * c.x.cp <= 32 * DH_KSC &&
                c.ldw >= (int64_t)c.k * c.x.cp && c.n_out > 0 && c.n_out % 8 == 0 && c.n_out <= 128 && DH_G * c.n_rows <= 32 * DH_RBW,
                "dec_head_fwd: transposed convolution (k %d, cp %d, n_rows %d, n_out %d) outside the fused kernel's limits", c.k, c.x.cp, c.n_rows, c.n_out);
    const int pitch = (int)(c.x.bstride / c.x.cp);
    C1V_REQUIRE(c.x.bstride % c.x.cp == 0 && c.n_rows + c.k - 1 <= pitch && ed.dst.off + ed.dst.len <= pitch, "dec_head_fwd: window overruns the sample pitch");
    const c1v_epilogue& ec = c.epi;
    C1V_REQUIRE(ec.row_mode == C1V_ROW_UP2 && ec.reshape_w <= 0 && ec.act.data == nullptr && ec.keep == nullptr && ec.sel == nullptr &&
                ec.dst.dtype == C1V_BF16 && ec.dst.data && ec.dst.cp >= c.n_out && ec.dst.cp % 8 == 0 && ec.dst.bstride % 8 == 0 &&
                ((uintptr_t)ec.dst.data % 16) == 0 && ec.dst.batch == l.B, "dec_head_fwd: the transposed convolution's epilogue must be a bf16 UP2 store");
    C1V_REQUIRE(((uintptr_t)d.w_op % 16) == 0 && ((uintptr_t)c.w_op % 16) == 0 && d.ldw % 8 == 0 && c.ldw % 8 == 0 && ((uintptr_t)c.x.data % 16) == 0 &&
                c.x.bstride % 8 == 0, "dec_head_fwd: operands must be 16-byte aligned");
    p = DecHeadDev{};
    p.lat = l;
    p.wd = (const bf16_raw*)d.w_op; p.ldwd = d.ldw; p.Kd = d.K; p.Nd = d.n_out; p.reshape_w = ed.reshape_w; p.bias_row_w = ed.bias_row_w;
    p.relu_d = ed.relu; p.bias_d = ed.bias;
    p.xd = (bf16_raw*)c.x.data; p.xd_bstride = c.x.bstride; p.xd_off = ed.dst.off; p.xd_len = ed.dst.len; p.xd_cp = c.x.cp; p.pitch = pitch;
    p.wc = (const bf16_raw*)c.w_op; p.ldwc = c.ldw; p.k = c.k; p.n_rows = c.n_rows; p.F = c.n_out; p.relu_c = ec.relu; p.bias_c = ec.bias;
    p.xf = (bf16_raw*)ec.dst.data; p.xf_bstride = ec.dst.bstride; p.xf_off = ec.dst.off; p.xf_len = ec.dst.len; p.xf_cp = ec.dst.cp;
    p.lds_img = 16 * p.Kd * 2;
    p.lds_park = p.lds_img + (int)c1v_cdiv((int64_t)DH_G * pitch * p.xd_cp * 2, 16) * 16;
    p.lds_kl = p.lds_park + (int)c1v_cdiv((int64_t)DH_G * p.n_rows * p.F * 2, 16) * 16;
    p.lds_total = p.lds_kl + DH_G * 4 * 64 * 4;
    C1V_REQUIRE(p.lds_total <= 64 * 1024, "dec_head_fwd: %d bytes of LDS", p.lds_total);
    return 0;
}
extern "C" int c1v_dec_head_fwd_check(const c1v_dec_head_args* a) {
    DecHeadDev p;
    return dec_head_fill(a, p);
}
extern "C" int c1v_dec_head_fwd(const c1v_dec_head_args* a, void* stream) {
    DecHeadDev p;
    if (dec_head_fill(a, p)) return -1;
    hipLaunchKernelGGL(dec_head_fwd_kernel, dim3((unsigned)c1v_cdiv(p.lat.B, DH_G)), dim3(64 * DH_W), p.lds_total, (hipStream_t)stream, p);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// column sums (bias gradients): grid (column blocks of 64, splits); each block sums its row range
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ buf, int64_t rows, int64_t ld, int C,
                                                     float* __restrict__ out, int64_t slab_stride, int splits) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int64_t per = c1v_cdiv(rows, splits);
    const int64_t r0 = (int64_t)blockIdx.y * per;
    int64_t r1 = r0 + per; if (r1 > rows) r1 = rows;
    float s = 0.f;
    if (c < C) for (int64_t r = r0 + rl; r < r1; r += 4) s += ElemTraits<T>::load(buf, r * ld + c);
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < C) out[(int64_t)blockIdx.y * slab_stride + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int c1v_colsum(const c1v_colsum_args* a, void* stream) {
    C1V_REQUIRE(a && a->buf && a->out, "colsum: null argument");
    C1V_REQUIRE(a->rows > 0 && a->C > 0 && a->ld >= a->C && a->splits > 0, "colsum: bad shape");
    dim3 grid((unsigned)c1v_cdiv(a->C, 64), (unsigned)a->splits);
    if (a->dtype == C1V_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)a->buf, a->rows, a->ld, a->C, a->out, a->slab_stride, a->splits);
    else hipLaunchKernelGGL(colsum_kernel<bf16_raw>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)a->buf, a->rows, a->ld, a->C, a->out, a->slab_stride, a->splits);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// gradient finalize: sum the wgrad / colsum slabs of every parameter segment into the flat gradient
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

struct LossRedDev { const float* xent_rows; const int* ids; const float* kl_b; int B, T, B_global; float* out3; float* xent_b; int pre; };

__global__ __launch_bounds__(256) void grad_finalize_kernel(const c1v_seg_desc* __restrict__ segs, int n_seg, float* __restrict__ grad,
                                                            float* __restrict__ sq_partial, int seg_blocks, LossRedDev lr) {
    // with loss scalars (c1v_grad_finalize_loss) block 0 computes them -- dispatched first, so its serial tail hides
    // behind the segment blocks -- and the segment blocks are numbered from 1
    const int first = (int)gridDim.x - seg_blocks;
    if ((int)blockIdx.x < first) {
        loss_reduce_body(lr.xent_rows, lr.ids, lr.kl_b, lr.B, lr.T, lr.B_global, lr.out3, lr.xent_b, lr.pre != 0);
        return;
    }
    const int blk = (int)blockIdx.x - first;
    __shared__ float red[4];
    const int si = find_desc(segs, n_seg, blk);
    const c1v_seg_desc d = segs[si];
    const int64_t base = (int64_t)(blk - d.first_block) * C1V_CHUNK;
    float sq = 0.f;
    // slabs, segment offsets and counts are multiples of 4 floats except for a tail: float4 body, scalar tail
    const bool vec = (d.count % 4 == 0) && (d.slab_stride % 4 == 0) && (d.rep_stride % 4 == 0) && (d.offset % 4 == 0) &&
                     (((uintptr_t)d.slab) % 16 == 0);
    constexpr int KI = C1V_CHUNK / 1024;               // float4 groups per thread
    if (vec) {
        // All KI groups of a thread advance together through the slabs, four slabs per round: 4 KI independent 16-byte
        // requests are in flight per round (the kernel is latency-bound: one group at a time meant KI x splits/4 serial
        // round trips for the segments with many wgrad splits).  Slab order is (split, rep): the sum is reproducible.
        int64_t idx[KI];
        bool in[KI];
        float4 g[KI];
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            idx[k] = base + (int64_t)(k * 256 + threadIdx.x) * 4;
            in[k] = idx[k] < d.count;
            if (!in[k]) idx[k] = 0;                    // clamped: loads stay unconditional, results are dropped
            g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (!d.frozen) {
            const int ns = d.splits * d.reps;
            int64_t off = 0;
            int sp = 0, r = 0;
            for (int s0 = 0; s0 < ns; s0 += 4) {
                int64_t o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[q] = off;
                    if (s0 + q + 1 < ns) {
                        ++r; off += d.rep_stride;
                        if (r == d.reps) { r = 0; ++sp; off = (int64_t)sp * d.slab_stride; }
                    }
                }
                float4 v[4][KI];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int k = 0; k < KI; ++k) v[q][k] = *(const float4*)(d.slab + o[q] + idx[k]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (s0 + q < ns) {
#pragma unroll
                        for (int k = 0; k < KI; ++k) { g[k].x += v[q][k].x; g[k].y += v[q][k].y; g[k].z += v[q][k].z; g[k].w += v[q][k].w; }
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < KI; ++k)
            if (in[k]) {
                *(float4*)(grad + d.offset + idx[k]) = g[k];
                sq += (g[k].x * g[k].x + g[k].y * g[k].y) + (g[k].z * g[k].z + g[k].w * g[k].w);
            }
    } else {
        // odd counts / unaligned slabs (the 302-wide output bias): one element per thread and pass, eight slabs requested
        // per round with clamped, unconditional loads (an element-by-element loop over 18 splits was one serial round trip
        // per slab and made this single block the longest of the launch); same (split, rep) summation order
        for (int64_t i = base + threadIdx.x; i < base + C1V_CHUNK && i < d.count; i += 256) {
            float g = 0.f;
            if (!d.frozen) {
                const int ns = d.splits * d.reps;
                for (int s0 = 0; s0 < ns; s0 += 8) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int sq_ = s0 + q < ns ? s0 + q : ns - 1;
                        const int sp = sq_ / d.reps, r = sq_ - sp * d.reps;
                        v[q] = d.slab[(int64_t)sp * d.slab_stride + (int64_t)r * d.rep_stride + i];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (s0 + q < ns) g += v[q];
                }
            }
            grad[d.offset + i] = g;
            sq += g * g;
        }
    }
    sq = block_sum_256(sq, red);
    if (threadIdx.x == 0 && sq_partial) sq_partial[blk] = sq;
}

extern "C" int c1v_grad_finalize(const c1v_seg_desc* segs, int32_t n_seg, int32_t total_blocks, float* grad, float* sq_partial, void* stream) {
    C1V_REQUIRE(segs && grad && n_seg > 0 && total_blocks > 0, "grad_finalize: bad argument");
    hipLaunchKernelGGL(grad_finalize_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, segs, n_seg, grad, sq_partial,
                       total_blocks, LossRedDev{});
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_grad_finalize_loss(const c1v_seg_desc* segs, int32_t n_seg, int32_t total_blocks, float* grad, float* sq_partial,
                                      const c1v_loss_reduce_args* l, void* stream) {
    C1V_REQUIRE(segs && grad && n_seg > 0 && total_blocks > 0, "grad_finalize_loss: bad argument");
    C1V_REQUIRE(l && l->xent_rows && l->ids && l->out3 && l->B > 0 && l->T > 0 && l->B_global > 0, "grad_finalize_loss: bad loss argument");
    C1V_REQUIRE(!l->pre_summed || l->xent_b, "grad_finalize_loss: pre_summed needs xent_b");
    const LossRedDev lr{l->xent_rows, l->ids, l->kl_b, l->B, l->T, l->B_global, l->out3, l->xent_b, l->pre_summed};
    hipLaunchKernelGGL(grad_finalize_kernel, dim3((unsigned)total_blocks + 1), dim3(256), 0, (hipStream_t)stream, segs, n_seg, grad, sq_partial,
                       total_blocks, lr);
    C1V_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ sq_partial) {
    __shared__ float red[4];
    float sq = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { const float g = x[i]; sq += g * g; }
    sq = block_sum_256(sq, red);
    if (threadIdx.x == 0) sq_partial[blockIdx.x] = sq;
}
extern "C" int c1v_sqnorm_partial(const float* x, int64_t n, float* sq_partial, int32_t n_partial, void* stream) {
    C1V_REQUIRE(x && sq_partial && n > 0 && n_partial > 0 && n_partial <= 2048, "sqnorm_partial: bad argument");
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3((unsigned)n_partial), dim3(256), 0, (hipStream_t)stream, x, n, sq_partial);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Nadam (Keras 2.2 `Nadam.get_updates`, reference :1329) with global-norm clipping (`clip_norm`)
// ------------------------------------------------------------------------------------------------------
// state (device): [0] = iterations (as double), [1] = m_schedule, [2] = lr
// schedule scalars of the NEXT update (one thread); sq < 0: leave sched[7] (the sum of squares) alone
__device__ __forceinline__ void nadam_schedule(double* __restrict__ state, float* __restrict__ sched, float sq,
                                               uint64_t* __restrict__ ctr, double beta1, double beta2, double decay) {
    const double t = state[0] + 1.0;
    const double mu_t = beta1 * (1.0 - 0.5 * pow(0.96, t * decay));
    const double mu_t1 = beta1 * (1.0 - 0.5 * pow(0.96, (t + 1.0) * decay));
    const double m_new = state[1] * mu_t;
    const double m_next = m_new * mu_t1;
    sched[0] = (float)state[2];
    sched[1] = (float)mu_t;
    sched[2] = (float)mu_t1;
    sched[3] = (float)(1.0 - m_new);
    sched[4] = (float)(1.0 - m_next);
    sched[5] = (float)(1.0 - pow(beta2, t));
    sched[6] = (float)t;
    if (sq >= 0.f) sched[7] = sq;
    state[0] = t;
    state[1] = m_new;
    if (ctr) ctr[0] += 1;
}
__global__ __launch_bounds__(256) void nadam_advance_kernel(double* __restrict__ state, float* __restrict__ sched,
                                                            const float* __restrict__ sq_partial, int n_partial,
                                                            uint64_t* __restrict__ ctr, double beta1, double beta2, double decay) {
    __shared__ float red[4];
    float sq = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) sq += sq_partial[i];
    sq = block_sum_256(sq, red);
    if (threadIdx.x != 0) return;
    nadam_schedule(state, sched, sq, ctr, beta1, beta2, decay);
}
extern "C" int c1v_nadam_advance(double* state, float* sched, const float* sq_partial, int32_t n_partial, uint64_t* ctr, void* stream) {
    C1V_REQUIRE(state && sched && sq_partial && n_partial > 0, "nadam_advance: bad argument");
    hipLaunchKernelGGL(nadam_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, state, sched, sq_partial, n_partial, ctr, 0.9, 0.999, 0.004);
    C1V_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void nadam_kernel(c1v_nadam_args a) {
    // sum of squares of the gradient: from sched[7] (c1v_nadam_advance), or -- a.sq_partial given -- summed here by every
    // block in the same fixed order (a few hundred values; saves the separate single-block launch before this kernel)
    float sumsq;
    if (a.sq_partial) {
        __shared__ float red[4];
        float sq = 0.f;
        for (int i = threadIdx.x; i < a.n_partial; i += 256) sq += a.sq_partial[i];
        sumsq = block_sum_256(sq, red);
    } else {
        sumsq = a.sched[7];
    }
    const float norm = sqrtf(sumsq) * a.grad_scale;
    float scale = a.grad_scale;
    if (a.clipnorm > 0.f && norm >= a.clipnorm) scale *= a.clipnorm / norm;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.norm_out) a.norm_out[0] = norm;
        if (a.sq_partial) a.sched_out7[0] = sumsq;
        if (a.ctr) a.ctr[0] += 1;                      // end of the step: the device-side RNG counter moves on
    }
    const float lr = a.sched[0], mu_t = a.sched[1], mu_t1 = a.sched[2], om_new = a.sched[3], om_next = a.sched[4], ob2t = a.sched[5];
    auto upd = [&](float gi, float& pi, float& mi, float& vi) {
        const float g = gi * scale;
        const float g_prime = g / om_new;
        const float m_t = a.beta1 * mi + (1.0f - a.beta1) * g;
        const float m_t_prime = m_t / om_next;
        const float v_t = a.beta2 * vi + (1.0f - a.beta2) * g * g;
        const float v_t_prime = v_t / ob2t;
        const float m_bar = (1.0f - mu_t) * g_prime + mu_t1 * m_t_prime;
        pi = pi - lr * m_bar / (sqrtf(v_t_prime) + a.eps);
        mi = m_t;
        vi = v_t;
    };
    // 16-byte words (the kernel is seven streams of HBM traffic and nothing else: 28 bytes per parameter), the same
    // arithmetic per element as the scalar tail
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t n4 = 0;
    if (((((uintptr_t)a.p) | ((uintptr_t)a.g) | ((uintptr_t)a.m) | ((uintptr_t)a.v)) & 15) == 0) {
        n4 = a.n >> 2;
        for (int64_t i = tid0; i < n4; i += stride) {
            const float4 g4 = ((const float4*)a.g)[i];
            float4 p4 = ((float4*)a.p)[i], m4 = ((float4*)a.m)[i], v4 = ((float4*)a.v)[i];
            upd(g4.x, p4.x, m4.x, v4.x); upd(g4.y, p4.y, m4.y, v4.y); upd(g4.z, p4.z, m4.z, v4.z); upd(g4.w, p4.w, m4.w, v4.w);
            ((float4*)a.p)[i] = p4; ((float4*)a.m)[i] = m4; ((float4*)a.v)[i] = v4;
        }
    }
    for (int64_t i = 4 * n4 + tid0; i < a.n; i += stride) upd(a.g[i], a.p[i], a.m[i], a.v[i]);
}
extern "C" int c1v_nadam_step(const c1v_nadam_args* a, void* stream) {
    C1V_REQUIRE(a && a->p && a->g && a->m && a->v && a->sched, "nadam_step: null argument");
    C1V_REQUIRE(a->n > 0, "nadam_step: bad size");
    C1V_REQUIRE(a->sq_partial == nullptr || (a->n_partial > 0 && a->sched_out7 != nullptr), "nadam_step: sq_partial needs n_partial and sched_out7");
    hipLaunchKernelGGL(nadam_kernel, dim3(grid_for(a->n)), dim3(256), 0, (hipStream_t)stream, *a);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// EOS truncation + renormalisation (reference :542-565): one wave per sample
// ------------------------------------------------------------------------------------------------------
template <typename F>
__global__ __launch_bounds__(256) void eos_truncate_kernel(const F* __restrict__ y, int64_t ldy, int B, int T_len, int V,
                                                           F* __restrict__ out, int* __restrict__ end_idx) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const F* yb = y + (int64_t)b * T_len * ldy;             // rows of V values, ldy apart
    F* ob = out + (int64_t)b * T_len * (V - 1);
    int end = T_len;
    for (int t = 0; t < T_len; ++t) {
        // np.argmax: first maximum wins
        F best = sizeof(F) == 8 ? (F)-1.7e308 : (F)-3.0e38; int bi = 0x7fffffff;
        for (int c = lane; c < V; c += 64) { const F v = yb[(int64_t)t * ldy + c]; if (v > best) { best = v; bi = c; } }
        for (int o = 32; o > 0; o >>= 1) {
            const F ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (bi == V - 1) { end = t > 1 ? t : 1; break; }
    }
    if (lane == 0 && end_idx) end_idx[b] = end;
    for (int t = 0; t < T_len; ++t) {
        if (t < end) {
            F ss = (F)0;
            for (int c = lane; c < V - 1; c += 64) { const F v = yb[(int64_t)t * ldy + c]; ss += v * v; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const F nrm = sqrt(ss);
            for (int c = lane; c < V - 1; c += 64) ob[(int64_t)t * (V - 1) + c] = yb[(int64_t)t * ldy + c] / nrm;
        } else {
            for (int c = lane; c < V - 1; c += 64) ob[(int64_t)t * (V - 1) + c] = (F)0;
        }
    }
}
extern "C" int c1v_eos_truncate_normalize_ld(const float* y, int64_t ldy, int32_t B, int32_t T, int32_t V, float* out, int32_t* end_idx,
                                             void* stream) {
    C1V_REQUIRE(y && out, "eos_truncate_normalize: null argument");
    C1V_REQUIRE(B > 0 && T > 0 && V > 1 && ldy >= V, "eos_truncate_normalize: bad shape (B=%d T=%d V=%d ldy=%lld)", B, T, V, (long long)ldy);
    hipLaunchKernelGGL(eos_truncate_kernel<float>, dim3((unsigned)c1v_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, y, ldy, B, T, V, out, end_idx);
    C1V_LAUNCH_CHECK();
    return 0;
}
extern "C" int c1v_eos_truncate_normalize(const float* y, int32_t B, int32_t T, int32_t V, float* out, int32_t* end_idx, void* stream) {
    return c1v_eos_truncate_normalize_ld(y, V, B, T, V, out, end_idx, stream);
}
extern "C" int c1v_eos_truncate_normalize_f64(const double* y, int32_t B, int32_t T, int32_t V, double* out, int32_t* end_idx, void* stream) {
    C1V_REQUIRE(y && out, "eos_truncate_normalize_f64: null argument");
    C1V_REQUIRE(B > 0 && T > 0 && V > 1, "eos_truncate_normalize_f64: bad shape");
    hipLaunchKernelGGL(eos_truncate_kernel<double>, dim3((unsigned)c1v_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, y, (int64_t)V, B, T, V, out, end_idx);
    C1V_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG: element e uses block (e >> 2), word (e & 3); counter = (blk_lo, blk_hi, stream_lo, stream_hi)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}

__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t sid, uint64_t offset,
                                                            const uint64_t* __restrict__ ctr, uint64_t ctr_mul) {
    if (ctr) offset += ctr[0] * ctr_mul;
    const uint64_t first_blk = offset >> 2;
    const uint64_t last_blk = (offset + n - 1) >> 2;
    for (uint64_t blk = first_blk + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; blk <= last_blk; blk += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 w = philox4x32_10(make_uint4((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)), (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
        float v[4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float u1 = ((float)(ws[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);
            const float u2 = (float)(ws[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
            const float r = sqrtf(-2.0f * logf(u1));
            const float ang = 6.283185307179586f * u2;
            v[2 * p] = r * cosf(ang); v[2 * p + 1] = r * sinf(ang);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t e = blk * 4 + k;
            if (e >= offset && e < offset + (uint64_t)n) out[e - offset] = v[k];
        }
    }
}
__global__ __launch_bounds__(256) void philox_bernoulli_kernel(uint8_t* __restrict__ out, int64_t n, uint32_t thr, uint64_t seed, uint64_t sid, uint64_t offset,
                                                               const uint64_t* __restrict__ ctr, uint64_t ctr_mul) {
    if (ctr) offset += ctr[0] * ctr_mul;
    const uint64_t first_blk = offset >> 2;
    const uint64_t last_blk = (offset + n - 1) >> 2;
    for (uint64_t blk = first_blk + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; blk <= last_blk; blk += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 w = philox4x32_10(make_uint4((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)), (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t e = blk * 4 + k;
            if (e >= offset && e < offset + (uint64_t)n) out[e - offset] = (ws[k] >> 8) < thr ? 1 : 0;
        }
    }
}
extern "C" int c1v_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t offset, const uint64_t* ctr, uint64_t ctr_mul, void* stream) {
    C1V_REQUIRE(out && n > 0, "philox_normal: bad argument");
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for(n / 4 + 2)), dim3(256), 0, (hipStream_t)stream, out, n, seed, stream_id, offset, ctr, ctr_mul);
    C1V_LAUNCH_CHECK();
    return 0;
}
extern "C" int c1v_philox_bernoulli(uint8_t* out, int64_t n, float keep_prob, uint64_t seed, uint64_t stream_id, uint64_t offset, const uint64_t* ctr, uint64_t ctr_mul, void* stream) {
    C1V_REQUIRE(out && n > 0 && keep_prob > 0.f && keep_prob <= 1.f, "philox_bernoulli: bad argument");
    const uint32_t thr = (uint32_t)(keep_prob * 16777216.0f);
    hipLaunchKernelGGL(philox_bernoulli_kernel, dim3(grid_for(n / 4 + 2)), dim3(256), 0, (hipStream_t)stream, out, n, thr, seed, stream_id, offset, ctr, ctr_mul);
    C1V_LAUNCH_CHECK();
    return 0;
}

// several RNG requests in one launch (one training step draws eps + two dropout masks)
__device__ __forceinline__ void philox_multi_body(const c1v_rng_desc* __restrict__ descs, int n_desc, uint64_t seed,
                                                  const uint64_t* __restrict__ ctr, int blkid) {
    const int di = find_desc(descs, n_desc, blkid);
    const c1v_rng_desc d = descs[di];
    const uint64_t offset = d.offset + (ctr ? ctr[0] * d.ctr_mul : 0);
    const uint64_t blk = (offset >> 2) + (uint64_t)(blkid - d.first_block) * 256 + threadIdx.x;
    if (blk > ((offset + d.n - 1) >> 2)) return;
    const uint4 w = philox4x32_10(make_uint4((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)d.stream_id, (uint32_t)(d.stream_id >> 32)),
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    if (d.kind == 0) {
        float v[4];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float u1 = ((float)(ws[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);
            const float u2 = (float)(ws[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
            const float r = sqrtf(-2.0f * logf(u1));
            const float ang = 6.283185307179586f * u2;
            v[2 * p] = r * cosf(ang); v[2 * p + 1] = r * sinf(ang);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t e = blk * 4 + k;
            if (e >= offset && e < offset + (uint64_t)d.n) ((float*)d.out)[e - offset] = v[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t e = blk * 4 + k;
            if (e >= offset && e < offset + (uint64_t)d.n) ((uint8_t*)d.out)[e - offset] = (ws[k] >> 8) < d.thr ? 1 : 0;
        }
    }
}
__global__ __launch_bounds__(256) void philox_multi_kernel(const c1v_rng_desc* __restrict__ descs, int n_desc, uint64_t seed,
                                                           const uint64_t* __restrict__ ctr) {
    philox_multi_body(descs, n_desc, seed, ctr, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------
// step prologue: everything a step needs before its first GEMM and that depends on nothing but the inputs -- the random
// draws, the encoder operand (pack or gather) and the live-row table of the loss -- in ONE launch (each was a 5-11 us
// launch of its own on the critical path of a captured step)
// ------------------------------------------------------------------------------------------------------
struct PrologueDev {
    const c1v_rng_desc* rng_descs; int n_rng, rng_blocks; uint64_t seed; const uint64_t* ctr;
    int in_kind, in_dtype, in_blocks;            // 0 none, 1 pack, 2 gather
    const float* src; int64_t ld; const int* in_ids; int T, C; c1v_seq dst;
    int live_blocks; const int* ids; int M, items, vocab_tiles, max_splits; int* pos; int* ctl; int* live_rows; int* live_cids;
    int sched_blocks; double* nadam_state; float* sched;
    int pack_blocks; const c1v_pack_desc* pack_descs; int n_pack;
};
__global__ __launch_bounds__(256) void step_prologue_kernel(const PrologueDev d) {
    int blk = (int)blockIdx.x;
    if (blk < d.live_blocks) {
        if (d.live_blocks > 1) live_rows_mb_body(d.ids, d.M, d.items, d.vocab_tiles, d.max_splits, d.pos, d.ctl, blk, d.live_blocks, d.live_rows, d.live_cids);
        else live_rows_body(d.ids, d.M, d.items, d.vocab_tiles, d.max_splits, d.pos, d.ctl, d.live_rows, d.live_cids);
        return;
    }
    blk -= d.live_blocks;
    if (blk < d.sched_blocks) {                    // Nadam schedule scalars of this step's update (c1v_nadam_advance without the norm)
        if (threadIdx.x == 0) nadam_schedule(d.nadam_state, d.sched, -1.0f, nullptr, 0.9, 0.999, 0.004);
        return;
    }
    blk -= d.sched_blocks;
    if (blk < d.rng_blocks) { philox_multi_body(d.rng_descs, d.n_rng, d.seed, d.ctr, blk); return; }
    blk -= d.rng_blocks;
    if (blk < d.pack_blocks) { pack_weights_body(d.pack_descs, d.n_pack, blk); return; }
    blk -= d.pack_blocks;
    if (d.in_kind == 1) {
        if (d.in_dtype == C1V_F32) pack_input_body<float>(d.src, d.T, d.C, d.dst, blk, d.in_blocks);
        else pack_input_body<bf16_raw>(d.src, d.T, d.C, d.dst, blk, d.in_blocks);
    } else if (d.in_kind == 2) {
        if (d.in_dtype == C1V_F32) gather_input_body<float>(d.src, d.ld, d.in_ids, d.T, d.C, d.dst, blk, d.in_blocks);
        else gather_input_body<bf16_raw>(d.src, d.ld, d.in_ids, d.T, d.C, d.dst, blk, d.in_blocks);
    }
}
static int check_seq(const c1v_seq& s, const char* what);
extern "C" int c1v_step_prologue(const c1v_step_prologue_args* a, void* stream) {
    C1V_REQUIRE(a, "step_prologue: null argument");
    C1V_REQUIRE(!(a->pack && a->gather), "step_prologue: pack and gather are alternatives");
    PrologueDev d{};
    if (a->n_rng > 0) {
        C1V_REQUIRE(a->rng_descs && a->rng_blocks > 0, "step_prologue: bad rng table");
        d.rng_descs = a->rng_descs; d.n_rng = a->n_rng; d.rng_blocks = a->rng_blocks; d.seed = a->seed; d.ctr = a->ctr;
    }
    if (a->pack || a->gather) {
        const c1v_seq& dst = a->pack ? a->pack->dst : a->gather->dst;
        const int T = a->pack ? a->pack->T : a->gather->T, C = a->pack ? a->pack->C : a->gather->C;
        if (check_seq(dst, "step_prologue dst")) return -1;
        C1V_REQUIRE(T > 0 && T <= dst.len && C > 0 && C <= dst.cp, "step_prologue: input shape mismatch (T=%d C=%d)", T, C);
        C1V_REQUIRE(dst.cp % 8 == 0 && dst.bstride % 8 == 0 && ((uintptr_t)dst.data % 32) == 0, "step_prologue: dst rows must be 8-element groups, 32-byte aligned");
        d.dst = dst; d.T = T; d.C = C; d.in_dtype = dst.dtype;
        if (a->pack) { C1V_REQUIRE(a->pack->src, "step_prologue: null input"); d.in_kind = 1; d.src = a->pack->src; }
        else {
            C1V_REQUIRE(a->gather->table && a->gather->ids && a->gather->ld >= C, "step_prologue: bad gather table");
            d.in_kind = 2; d.src = a->gather->table; d.ld = a->gather->ld; d.in_ids = a->gather->ids;
        }
        d.in_blocks = (int)grid_for((int64_t)dst.batch * T * (dst.cp / 8));
    }
    if (a->live) {
        const c1v_live_rows_args* l = a->live;
        C1V_REQUIRE(l->ids && l->pos && l->ctl && l->M > 0 && l->items > 0 && l->vocab_tiles > 0 && l->max_splits > 0, "step_prologue: bad live-row argument");
        C1V_REQUIRE(l->items >= (int)c1v_cdiv(l->M, VS_BM_ROWS), "step_prologue: the stream grid must cover all row tiles");
        // (1 024 rows per 256-thread workgroup, up to LIVE_MB_MAX of them; larger tables stay with one workgroup's walk --
        // and large batches build theirs in c1v_live_rows_gather anyway)
        const char* env_mb = getenv("C1V_LIVE_MB");
        const bool mb_off = env_mb && env_mb[0] == '0';
        const int nb = (int)c1v_cdiv(l->M, 1024);
        d.live_blocks = (nb <= LIVE_MB_MAX && !mb_off) ? nb : 1; d.ids = l->ids; d.M = l->M; d.items = l->items; d.vocab_tiles = l->vocab_tiles; d.max_splits = l->max_splits;
        d.pos = l->pos; d.ctl = l->ctl;
        C1V_REQUIRE((l->rows != nullptr) == (l->cids != nullptr), "step_prologue: live rows / cids come together");
        d.live_rows = l->rows; d.live_cids = l->cids;
    }
    if (a->nadam_state) {
        C1V_REQUIRE(a->sched, "step_prologue: nadam_state needs sched");
        d.sched_blocks = 1; d.nadam_state = a->nadam_state; d.sched = a->sched;
    }
    if (a->n_pack > 0) {
        C1V_REQUIRE(a->pack_descs && a->pack_blocks > 0, "step_prologue: bad weight-pack table");
        d.pack_blocks = a->pack_blocks; d.pack_descs = a->pack_descs; d.n_pack = a->n_pack;
    }
    const int total = d.live_blocks + d.sched_blocks + d.rng_blocks + d.pack_blocks + d.in_blocks;
    C1V_REQUIRE(total > 0, "step_prologue: nothing to do");
    hipLaunchKernelGGL(step_prologue_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, d);
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_philox_multi(const c1v_rng_desc* descs, int32_t n_desc, int32_t total_blocks, uint64_t seed, const uint64_t* ctr, void* stream) {
    C1V_REQUIRE(descs && n_desc > 0 && total_blocks > 0, "philox_multi: bad argument");
    hipLaunchKernelGGL(philox_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, descs, n_desc, seed, ctr);
    C1V_LAUNCH_CHECK();
    return 0;
}

// Large batches (see VAEEngine._input_own_launch): the live-row table and the encoder operand in ONE launch of 1024-thread
// blocks -- block 0 scans the targets with 16 waves, the others assemble the input; neither needs the other's result.
struct LiveGatherDev {
    const int* ids; int M, items, vocab_tiles, max_splits; int* pos; int* ctl; int* rows; int* cids; int live_blocks;
    const float* table; int64_t ld; const int* in_ids; int T, C; c1v_seq dst; int in_blocks, in_dtype;
};
__global__ __launch_bounds__(1024) void live_gather_kernel(const LiveGatherDev d) {
    // the live-row table: several workgroups of 4 096 rows, each counting the rows in front of its range itself
    // (live_rows_mb_body; one workgroup walking C4's 270 k rows took 43 us, 120 with the inverse table's scattered stores
    // -- longer than the gather it shares the launch with)
    if ((int)blockIdx.x < d.live_blocks) {
        if (d.live_blocks > 1) live_rows_mb_body(d.ids, d.M, d.items, d.vocab_tiles, d.max_splits, d.pos, d.ctl, (int)blockIdx.x, d.live_blocks, d.rows, d.cids);
        else live_rows_body(d.ids, d.M, d.items, d.vocab_tiles, d.max_splits, d.pos, d.ctl, d.rows, d.cids);
        return;
    }
    const int blk = (int)blockIdx.x - d.live_blocks;
    if (d.in_dtype == C1V_F32) gather_input_body<float>(d.table, d.ld, d.in_ids, d.T, d.C, d.dst, blk, d.in_blocks);
    else gather_input_body<bf16_raw>(d.table, d.ld, d.in_ids, d.T, d.C, d.dst, blk, d.in_blocks);
}
extern "C" int c1v_live_rows_gather(const c1v_live_rows_args* l, const c1v_gather_input_args* a, void* stream) {
    C1V_REQUIRE(l && a, "live_rows_gather: null argument");
    C1V_REQUIRE(l->ids && l->pos && l->ctl && l->M > 0 && l->items > 0 && l->vocab_tiles > 0 && l->max_splits > 0, "live_rows_gather: bad live-row argument");
    C1V_REQUIRE(l->items >= (int)c1v_cdiv(l->M, VS_BM_ROWS), "live_rows_gather: the stream grid must cover all row tiles");
    C1V_REQUIRE(a->table && a->ids, "live_rows_gather: null gather argument");
    if (check_seq(a->dst, "live_rows_gather dst")) return -1;
    C1V_REQUIRE(a->T > 0 && a->T <= a->dst.len && a->C > 0 && a->C <= a->dst.cp && a->ld >= a->C, "live_rows_gather: shape mismatch");
    C1V_REQUIRE(a->dst.cp % 8 == 0 && a->dst.bstride % 8 == 0 && ((uintptr_t)a->dst.data % 32) == 0, "live_rows_gather: dst rows must be 8-element groups, 32-byte aligned");
    LiveGatherDev d{};
    d.ids = l->ids; d.M = l->M; d.items = l->items; d.vocab_tiles = l->vocab_tiles; d.max_splits = l->max_splits; d.pos = l->pos; d.ctl = l->ctl;
    C1V_REQUIRE((l->rows != nullptr) == (l->cids != nullptr), "live_rows_gather: live rows / cids come together");
    d.rows = l->rows; d.cids = l->cids;
    d.table = a->table; d.ld = a->ld; d.in_ids = a->ids; d.T = a->T; d.C = a->C; d.dst = a->dst; d.in_dtype = a->dst.dtype;
    const int64_t total = (int64_t)a->dst.batch * a->T * (a->dst.cp / 8);
    int64_t nb = c1v_cdiv(total, 1024 * 4);                        // ~4 eight-element groups per thread
    d.in_blocks = (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
    const int lb = (int)c1v_cdiv(l->M, 4096);
    const char* env_mb = getenv("C1V_LIVE_MB");
    d.live_blocks = (lb <= 256 && !(env_mb && env_mb[0] == '0')) ? lb : 1;
    hipLaunchKernelGGL(live_gather_kernel, dim3((unsigned)(d.live_blocks + d.in_blocks)), dim3(1024), 0, (hipStream_t)stream, d);
    C1V_LAUNCH_CHECK();
    return 0;
}

// device-side step counter used by captured graphs (RNG offsets advance without host involvement)
__global__ void counter_add_kernel(uint64_t* ctr, uint64_t inc) { if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += inc; }
extern "C" int c1v_counter_add(uint64_t* ctr, uint64_t inc, void* stream) {
    C1V_REQUIRE(ctr, "counter_add: null argument");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctr, inc);
    C1V_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void loss_sample_means_kernel(const float* __restrict__ xent_rows, const int* __restrict__ ids, int B, int T_len,
                                                                float* __restrict__ xent_b) {
    loss_sample_means_body(xent_rows, ids, B, T_len, xent_b, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
}
extern "C" int c1v_loss_sample_means(const c1v_loss_reduce_args* l, void* stream) {
    C1V_REQUIRE(l && l->xent_rows && l->ids && l->xent_b && l->B > 0 && l->T > 0, "loss_sample_means: bad argument");
    hipLaunchKernelGGL(loss_sample_means_kernel, dim3((unsigned)c1v_cdiv(l->B, 4)), dim3(256), 0, (hipStream_t)stream, l->xent_rows, l->ids, l->B, l->T, l->xent_b);
    C1V_LAUNCH_CHECK();
    return 0;
}

extern "C" int c1v_grad_finalize_step(const c1v_grad_finalize_args* a, void* stream) {
    C1V_REQUIRE(a, "grad_finalize_step: null argument");
    if (a->loss) return c1v_grad_finalize_loss(a->segs, a->n_seg, a->total_blocks, a->grad, a->sq_partial, a->loss, stream);
    return c1v_grad_finalize(a->segs, a->n_seg, a->total_blocks, a->grad, a->sq_partial, stream);
}
