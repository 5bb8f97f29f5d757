#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
// MODE 0: MFMA only; 1: + 2 simple VALU (independent) per gap; 2: + 1 trans (v_exp) per gap; 3: + 1 trans + 2 simple; 4: only the VALU (no MFMA): 1 trans + 2 simple
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = lane * 0.01f, b = lane * 0.02f;
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = lane * 0.001f * (i + 1); y[i] = 0.5f + i; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE != 4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 1 || MODE == 3 || MODE == 4) { asm volatile("v_mul_f32 %0, 0x3f7fff00, %0" : "+v"(x[i])); asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(y[(i + 3) & 7])); }
            if (MODE == 2 || MODE == 3 || MODE == 4) { asm volatile("v_exp_f32 %0, %0" : "+v"(y[(i + 5) & 7])); }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + x[i] + y[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.2f cycles per MFMA slot\n", name, (double)c / (iters * 8.0));
}
int main() {
    run<0>("mfma only"); run<1>("mfma + 2 simple VALU"); run<2>("mfma + 1 trans"); run<3>("mfma + 1 trans + 2 simple"); run<4>("no mfma: 1 trans + 2 simple");
    return 0;
}
