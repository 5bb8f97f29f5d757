// Do fp32 MFMAs of one wave overlap with VALU work of ANOTHER wave on the same SIMD (gfx950)?
// 512-thread blocks = 8 waves = 2 per SIMD (waves w and w + 4 share SIMD w % 4).  Waves 0-3 run an MFMA chain, waves 4-7
// run a VALU chain (or exit), each timing itself with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_2wave.hip -o mfma_valu_2wave && ./mfma_valu_2wave
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

// MODE 0: MFMA waves only; 1: VALU waves only; 2: both; 3: MFMA on all 8 waves; 4: VALU on all 8 waves; 5: bf16 MFMA waves + VALU waves
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool mfma_wave = MODE == 3 ? true : (MODE == 4 ? false : wave < 4);
    const bool active = (MODE == 0) ? wave < 4 : (MODE == 1 ? wave >= 4 : true);
    float s = 0.f;
    long long t0 = 0, t1 = 0;
    if (active) {
        if (mfma_wave) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
            float a = lane * 0.01f, b = lane * 0.02f;
            t0 = clock64();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == 5) {
                        using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
                        bf16x8 av, bv;
                        for (int q = 0; q < 8; ++q) { av[q] = (__bf16)a; bv[q] = (__bf16)b; }
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
                    } else {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                    }
                }
            }
            t1 = clock64();
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        } else {
            float x[8], y[8];
            for (int i = 0; i < 8; ++i) { x[i] = lane * 0.001f * (i + 1); y[i] = 0.5f + i; }
            t0 = clock64();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_mul_f32 %0, 0x3f7fff00, %0" : "+v"(x[i]));
                    asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(y[(i + 3) & 7]));
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + 1) & 7]) : "v"(y[i]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(y[(i + 5) & 7]));
                }
            }
            t1 = clock64();
            for (int i = 0; i < 8; ++i) s += x[i] + y[i];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    hipMemset(cyc, 0, 64);
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-46s mfma wave0: %7.2f cyc / MFMA   valu wave4: %7.2f cyc / 4-op group   (wave3 %.2f, wave7 %.2f)\n", name,
           (double)c[0] / (iters * 8.0), (double)c[4] / (iters * 8.0), (double)c[3] / (iters * 8.0), (double)c[7] / (iters * 8.0));
}
int main() {
    run<0>("fp32 MFMA waves alone (1 per SIMD)");
    run<1>("VALU waves alone (1 per SIMD)");
    run<2>("fp32 MFMA wave + VALU wave per SIMD");
    run<3>("fp32 MFMA on both waves of a SIMD");
    run<4>("VALU on both waves of a SIMD");
    run<5>("bf16 16x16x32 MFMA wave + VALU wave per SIMD");
    return 0;
}
