#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4_t __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
// MODE 0: ds_read_b128, 1: ds_read_b64, 2: ds_read_b64_tr_b16
template <int MODE>
__global__ __launch_bounds__(256) void rd(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 256) ((float*)smem)[i] = (float)i;
    __syncthreads();
    unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    const char* base = smem + wave * 4096;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int off = ((i * 8 + u) & 7) * 1024; asm volatile("" : "+v"(off));
            const char* p = base + off;
            if (MODE == 0) {
                uint4 v = *(const uint4*)(p + lane * 16);
                acc0 += v.x; acc1 += v.y; acc2 += v.z; acc3 += v.w;
            } else if (MODE == 1) {
                uint2 v = *(const uint2*)(p + lane * 8);
                acc0 += v.x; acc1 += v.y;
            } else {
                s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p + lane * 8));
                uint2 w = __builtin_bit_cast(uint2, v);
                acc0 += w.x; acc1 += w.y;
            }
        }
    }
    out[blockIdx.x * 256 + tid] = (float)(acc0 + acc1 + acc2 + acc3);
}
template <int MODE> void run(float* d, const char* name, int bytes_per_lane) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rd<MODE><<<256, 256, 65536>>>(d, 10);
    (void)hipEventRecord(e0);
    rd<MODE><<<256, 256, 65536>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double bytes_cu = (double)iters * 8 * 256 * bytes_per_lane;      // per CU (one block per CU)
    printf("%s: %.1f B/ns per CU = %.1f B/clk @2.4GHz (%.2f ms)\n", name, bytes_cu / (ms * 1e6), bytes_cu / (ms * 1e6) / 2.4, ms);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 256 * 4);
    (void)hipFuncSetAttribute((const void*)rd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)rd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)rd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0>(d, "ds_read_b128", 16); run<1>(d, "ds_read_b64", 8); run<2>(d, "ds_read_b64_tr_b16", 8);
    return 0;
}
