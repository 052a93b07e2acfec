// Shared device/host helpers of libc1dvae (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include "../../include/c1dvae.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef unsigned short bf16_raw;

#define C1V_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_raw h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(bf16_raw, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kDtype = C1V_F32;
    static constexpr int kPer16B = 4;
    __device__ static __forceinline__ float load(const void* p, int64_t i) { return ((const float*)p)[i]; }
    __device__ static __forceinline__ void store(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};
template <> struct ElemTraits<bf16_raw> {
    static constexpr int kDtype = C1V_BF16;
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float load(const void* p, int64_t i) { return bf16_to_f32(((const bf16_raw*)p)[i]); }
    __device__ static __forceinline__ void store(void* p, int64_t i, float v) { ((bf16_raw*)p)[i] = f32_to_bf16(v); }
};

__device__ __forceinline__ float load_any(const void* p, int dtype, int64_t i) {
    return dtype == C1V_F32 ? ((const float*)p)[i] : bf16_to_f32(((const bf16_raw*)p)[i]);
}
__device__ __forceinline__ void store_any(void* p, int dtype, int64_t i, float v) {
    if (dtype == C1V_F32) ((float*)p)[i] = v; else ((bf16_raw*)p)[i] = f32_to_bf16(v);
}

// LDS-DMA (global -> LDS, 16 bytes per lane, lane-linear: lane L lands at lds_addr + 16 L) issued through inline asm ON
// PURPOSE: for the builtin, the compiler cannot tell which LDS bytes a request in flight will write, so it puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read -- the request just issued for a LATER stage is waited for at once
// and a multi-stage ring degenerates into one exposed memory latency per K step (measured: 0.9 us per step).  Through
// asm the request is invisible to that bookkeeping; the kernels count their own `s_waitcnt vmcnt(N)` + barrier before a
// stage is read.  `lds_addr` is wave-uniform (LDS byte address of the wave's 1 KiB piece).
__device__ __forceinline__ uint32_t c1v_lds_addr(const void* p) { return (uint32_t)(uintptr_t)C1V_LDS_PTR(const void, p); }
__device__ __forceinline__ void c1v_glds16(const void* gsrc, uint32_t lds_addr) {
    const uint32_t a = __builtin_amdgcn_readfirstlane(lds_addr);      // an "s" operand must really live in an SGPR
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(a), "v"(gsrc) : "memory", "m0");
}

// Same request with the address split into a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit
// VALU addition per request when only the base changes from one request to the next.
__device__ __forceinline__ void c1v_glds16_so(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    const uint32_t a = __builtin_amdgcn_readfirstlane(lds_addr);
    const uint64_t b = (uint64_t)(uintptr_t)sbase;
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const uint64_t bs = ((uint64_t)bhi << 32) | blo;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(a), "v"(voff), "s"(bs) : "memory", "m0");
}

// Touch the 64-byte lines [1, LINES] of the kernel-argument segment with independent scalar loads and wait once: the
// compiler fetches by-value argument structs lazily, one cold line (a trip to memory, ~1k cycles) at a time, wherever a
// field is first used; after this every such fetch hits the scalar cache.  (HIP appends its hidden arguments to the
// segment, so a touch up to 256 bytes past the explicit arguments is still inside it.)
template <int LINES> __device__ __forceinline__ void c1v_touch_kernarg() {
    static_assert(LINES >= 1 && LINES <= 7, "1..7 lines");
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t d0, d1, d2, d3, d4, d5, d6;
    if constexpr (LINES <= 4)
        asm volatile("s_load_dword %0, %4, 0x40\n\ts_load_dword %1, %4, 0x80\n\ts_load_dword %2, %4, 0xc0\n\ts_load_dword %3, %4, 0x100\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka) : "memory");
    else
        asm volatile("s_load_dword %0, %7, 0x40\n\ts_load_dword %1, %7, 0x80\n\ts_load_dword %2, %7, 0xc0\n\ts_load_dword %3, %7, 0x100\n\t"
                     "s_load_dword %4, %7, 0x140\n\ts_load_dword %5, %7, 0x180\n\ts_load_dword %6, %7, 0x1c0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6) : "s"(ka) : "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The same sum through DPP row operations and four lane reads instead of six LDS-crossbar permutes (each of those is an LDS
// round trip; a wave that is alone on its dependency chain waits for every one).  Another order of the additions than
// wave_sum: not interchangeable where results are compared to the bit.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
}

// ---- host side -------------------------------------------------------------------------------------
void c1v_set_error(const char* fmt, ...);
#define C1V_REQUIRE(cond, ...) do { if (!(cond)) { c1v_set_error(__VA_ARGS__); return -1; } } while (0)
#define C1V_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
        c1v_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return -2; } } while (0)

static inline int c1v_esize(int dtype) { return dtype == C1V_F32 ? 4 : 2; }
__host__ __device__ static inline int64_t c1v_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// device-side copy of c1v_epilogue plus the row enumeration of the launch
struct EpiDev {
    const float* bias; int relu; int row_mode; int reshape_w; int exp2;
    void* dst; int dst_dtype; int64_t dst_bstride; int dst_off, dst_len, dst_cp, dst_c;
    const void* act; int act_dtype; int64_t act_bstride; int act_off, act_cp, act_rowmul; float act_scale;
    uint8_t* sel; int64_t sel_bstride; int sel_cp;
    const uint8_t* keep; float keep_scale;
    int M, Le, nrows, N;
    const float* lat_mlv; const float* lat_eps; const uint8_t* lat_keep; int lat_D; float lat_keep_prob, lat_std_scale, lat_kc;
    int bias_row_w;       // c1v_epilogue::bias_row_w
    void* nrm_yhat; int nrm_kp; float* nrm_rnorm; float nrm_qscale; const int* nrm_pos; int nrm_live_only; int nrm_T;   // C1V_ROW_NORM (nrm_T = compute dtype)
    char* nrm_dy; int64_t nrm_dy_bstride_b; int nrm_dy_off, nrm_dy_row_b;      // rows without a target: zero row (b, u) of this buffer (byte strides)
    int fastw;            // 4 / 2: every group address of the epilogue is a multiple of that many elements (vector path); 0: element-wise
};

struct NtParams {
    const char* A; int64_t a_bstride_b; int64_t lda_b; int a_off;   // A row(m): b=m/Le,u=m%Le -> A + b*a_bstride_b + (a_off+u)*lda_b
    const char* W; int64_t ldw_b;
    int64_t Kb;                                                      // K in bytes
    int ksplits, kt_per_split; int64_t dst_split_stride;             // split-K (grid z): K tiles [s*kt_per_split, ..) -> dst + s*stride
    // Order of the K tiles of a convolution (k_taps > 1): K = taps x (k_strips 128-byte column strips of a buffer row).  Tile
    // t covers strip t / k_taps of tap t % k_taps, i.e. byte offset (t % k_taps) * k_strips * 128 + (t / k_taps) * 128 of
    // both operands' rows: the taps of one strip are consecutive tiles, and tap j + 1 re-reads the lines tap j has just
    // fetched (the same buffer rows, shifted by one) while they are still in the XCD's L2.  k_taps <= 1: natural order.
    int k_taps, k_strips;
    int wide;                                                         // ntw_body over the flat buffer rows (e.Le = sample pitch): 1 = 128 x 128 (two workgroups per CU), 2 = 256 x 128
    EpiDev e;
};

struct TnParams {
    const char* A; int64_t lda_b; int I; int Ia;     // Ia = readable columns (multiple of 16 bytes)
    const char* D; int64_t ldd_b; int J; int Ja;
    int64_t r_begin, r_end; int64_t rows_per_split;
    int a_shift0; int taps; int flip; int64_t tap_stride, si, sj;
    float* out; int64_t slab_stride; int splits;
    float* bias_out; int64_t bias_slab_stride;
    int tg;                                          // 0: one workgroup per tap (tn_body); 2 / 3: taps fused (tn_taps_body)
};

int c1v_launch_nt(int dtype, const NtParams& p, hipStream_t st);
int c1v_launch_tn(int dtype, const TnParams& p, hipStream_t st);
constexpr int C1V_TN_MULTI_MAX = 12;
int c1v_launch_tn_multi(int dtype, const TnParams* ps, int n, hipStream_t st);
int c1v_launch_pair(int dtype, const NtParams& nt, const TnParams& tn, hipStream_t st);
int c1v_check_nt(const NtParams& p);
int c1v_check_tn(const TnParams& p);
int c1v_fill_epi(const c1v_epilogue& e, EpiDev& d, int M, int Le, int nrows, int N);
