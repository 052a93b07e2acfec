// v_mfma_f32_32x32x16_bf16 issue rate of one (or two) waves per SIMD with K independent accumulator chains and VALU
// instructions between the MFMAs -- in MEASURED shader cycles (s_memtime deltas of wave 0 of block 0) and in ns (HIP events),
// so the clock the part actually ran at is part of the result, on one workgroup per XCD (no power limit) and on all 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
// VK: 0 = v_fma_f32 (full rate), 1 = v_exp_f32 (transcendental, quarter rate), 2 = v_cvt_pk_bf16_f32
template <int K, int VALU, int VK>
__global__ __launch_bounds__(512) void chain(float* out, unsigned long long* cyc, int iters, float seed) {
    f32x16_t acc[K];
    for (int k = 0; k < K; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f); b[j] = (__bf16)(seed * 0.5f); }
    float vv[8]; for (int q = 0; q < 8; ++q) vv[q] = seed + q;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < VALU; ++q) {
                if constexpr (VK == 0) vv[q] = vv[q] * 1.0001f + 0.5f;
                else if constexpr (VK == 1) vv[q] = __builtin_amdgcn_exp2f(vv[q]);
                else { typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf2; bf2 p; p[0] = (__bf16)vv[q]; p[1] = (__bf16)vv[(q + 1) & 7];
                       vv[q] = __builtin_bit_cast(float, p); }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f; for (int q = 0; q < 8; ++q) s += vv[q];
    for (int k = 0; k < K; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int K, int VALU, int VK> void run(float* d, unsigned long long* dc, int grid, int threads) {
    const int iters = 20000;
    static const char* vk[] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<K, VALU, VK><<<grid, threads>>>(d, dc, 2000, 1.0f);
    hipEventRecord(e0);
    chain<K, VALU, VK><<<grid, threads>>>(d, dc, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * K, per_ns = ms * 1e6 / n / (threads / 256), per_tick = (double)c / n / (threads / 256);
    printf("grid %3d waves/SIMD %d chains %d  %d x %-14s per MFMA: %6.2f ns  %6.2f s_memtime ticks  (%.3f ticks/ns)\n", grid, threads / 256, K, VALU, VK < 3 ? vk[VK] : "",
           per_ns, per_tick, per_tick / per_ns);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    unsigned long long* dc; hipMalloc(&dc, 8);
    for (int grid : {8, 256}) {
        run<1, 0, 0>(d, dc, grid, 256); run<2, 0, 0>(d, dc, grid, 256); run<4, 0, 0>(d, dc, grid, 256);
        run<2, 2, 0>(d, dc, grid, 256); run<2, 4, 0>(d, dc, grid, 256); run<2, 7, 0>(d, dc, grid, 256); run<4, 4, 0>(d, dc, grid, 256);
        run<2, 1, 1>(d, dc, grid, 256); run<2, 2, 1>(d, dc, grid, 256); run<4, 1, 1>(d, dc, grid, 256);
        run<2, 2, 2>(d, dc, grid, 256);
        run<2, 0, 0>(d, dc, grid, 512); run<2, 4, 0>(d, dc, grid, 512); run<2, 2, 1>(d, dc, grid, 512);
    }
    return 0;
}
