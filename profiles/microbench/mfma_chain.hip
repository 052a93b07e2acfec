#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
template <int K, int VALU>
__global__ __launch_bounds__(512) void chain(float* out, int iters, float seed) {
    f32x16_t acc[K];
    for (int k = 0; k < K; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f); b[j] = (__bf16)(seed * 0.5f); }
    float vv[8]; for (int q = 0; q < 8; ++q) vv[q] = seed + q;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < VALU; ++q) vv[q] = vv[q] * 1.0001f + 0.5f;
        }
    }
    float s = 0.f; for (int q = 0; q < 8; ++q) s += vv[q];
    for (int k = 0; k < K; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int K, int VALU> void run(float* d, const char* name, int threads = 256) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<K, VALU><<<256, threads>>>(d, 100, 1.0f);
    hipEventRecord(e0);
    chain<K, VALU><<<256, threads>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per = ms * 1e-3 / ((double)iters * K * (threads / 256));
    printf("[thr %d] %s chains %d valu/mfma %d: %.1f ns per MFMA = %.1f cycles @2.4GHz, %.0f TF/s chip\n", threads, name, K, VALU, per * 1e9, per * 2.4e9,
           32768.0 * 1024 / per / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<1, 0>(d, "dep"); run<2, 0>(d, "dep"); run<4, 0>(d, "dep"); run<8, 0>(d, "dep");
    run<1, 4>(d, "dep+valu"); run<2, 4>(d, "dep+valu"); run<4, 4>(d, "dep+valu"); run<2, 7>(d, "dep+valu"); run<4, 7>(d, "dep+valu");
    run<2, 0>(d, "2waves", 512); run<2, 4>(d, "2waves", 512); run<2, 7>(d, "2waves", 512); run<1, 0>(d, "2waves", 512); run<1, 7>(d, "2waves", 512);
    return 0;
}
