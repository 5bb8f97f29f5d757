// What keeps the rollout kernel's k loop above 32 cycles per v_mfma_f32_16x16x4_f32?  (round 3: 36.5 cycles per MFMA inside
// wave_gemm<3,3,1> by the in-kernel phase profile, with buffer-load weight fragments and immediate-offset LDS reads.)
// The loop body of rollout.hpp wave_gemm<R=3,CT=3,EX=1> rebuilt stand-alone -- 10 accumulators, per 16-wide k chunk 4 weight
// fragments (buffer_load_dwordx4 ... s_off offen from an L2-resident array) + 4 activation fragments (ds_read_b128) feeding
// 4 k-steps x 10 MFMAs, double buffered, sched_barrier pinned like the original -- with its ingredients switchable:
//   bit 0: s_nop 1 in front of every k-step      bit 1: the LDS fragment reads      bit 2: the weight fragment loads
//   bit 3: weights from a per-workgroup private region (no sharing in L2 between workgroups) instead of one 585 KB member block
//   bit 4: the 8 loads of the NEXT chunk are not issued in one clump ahead of the 40 MFMAs but one after every 5th MFMA
//          (sched_barrier on both sides of each), i.e. inside the 32-cycle shadows of the MFMAs
// One workgroup = 4 waves = one per SIMD, 220 workgroups (cfg2's grid).  Prints cycles per MFMA (s_memtime of wave 0, workgroup 0)
// and the wall time per launch.   hipcc --offload-arch=gfx950 -O3 kloop_probe.hip -o kloop_probe && ./kloop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ void mfma(const float a, const float b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

constexpr int kKC = 13, kLayers = 8, kLd = 264, kRows = 48;

template <int MASK>
__global__ __launch_bounds__(256, 1) void probe(const float* w, float* out, long long* cyc, int reps, long long wg_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* act = reinterpret_cast<float*>(smem);  // [48][264]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kRows * kLd; i += 256) act[i] = 0.001f * (float)(i % 97);
    __syncthreads();
    constexpr bool NOP = MASK & 1, LDS = MASK & 2, VMEM = MASK & 4, PRIV = MASK & 8, INTER = MASK & 16;
    const float* wbase = w + (PRIV ? (long long)blockIdx.x * wg_stride : 0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), 0, 0x7FFFFFFF, 0x00020000);
    f32x4 acc[10];
    for (int i = 0; i < 10; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned woff[4];
    for (int ct = 0; ct < 4; ++ct) woff[ct] = (unsigned)(((wave + 4 * ct) * kKC * 64 + lane) * 16);
    const char* ap = reinterpret_cast<const char*>(act + (lane & 15) * kLd + 4 * (lane >> 4));
    struct Fr { f32x4 b[4], a[4]; };
    auto load = [&](Fr& f, const int layer, const int kk) __attribute__((always_inline)) {
        if constexpr (VMEM) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)woff[ct], layer * (13 * kKC * 1024) + kk * 1024, 0);
                __builtin_memcpy(&f.b[ct], &v, 16);
            }
        }
        if constexpr (LDS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) f.a[r] = *reinterpret_cast<const f32x4*>(ap + (r % 3) * 16 * kLd * 4 + kk * 64);
        }
    };
    auto compute = [&](const Fr& f) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (NOP) asm volatile("s_nop 1");
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int r = 0; r < 3; ++r) mfma(f.b[ct][s], f.a[r][s], acc[ct * 3 + r]);
            mfma(f.b[3][s], f.a[3][s], acc[9]);
        }
    };
    // one of the 8 loads of chunk kk (i < 4: weight fragment i, else activation fragment i - 4), pinned where it is written
    auto load_one = [&](Fr& f, const int layer, const int kk, const int i) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        if (i < 4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)woff[i], layer * (13 * kKC * 1024) + kk * 1024, 0);
            __builtin_memcpy(&f.b[i], &v, 16);
        } else {
            f.a[i - 4] = *reinterpret_cast<const f32x4*>(ap + ((i - 4) % 3) * 16 * kLd * 4 + kk * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // compute chunk `f` while the loads of chunk kk_next trickle into `g`: one load behind every 5th MFMA
    auto compute_interleaved = [&](const Fr& f, Fr& g, const int layer, const int kk_next) __attribute__((always_inline)) {
        int n = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (NOP) asm volatile("s_nop 1");
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                if (u < 9) mfma(f.b[u / 3][s], f.a[u % 3][s], acc[u]);
                else mfma(f.b[3][s], f.a[3][s], acc[9]);
                ++n;
                if (n % 5 == 0) load_one(g, layer, kk_next, n / 5 - 1);
            }
        }
    };
    Fr f0, f1;
    for (int i = 0; i < 4; ++i) { f0.b[i] = f1.b[i] = f32x4{1.f, 2.f, 3.f, 4.f}; f0.a[i] = f1.a[i] = f32x4{0.5f, 0.25f, 0.125f, 1.f}; }
    long long t0 = 0, t1 = 0, mf = 0;
    for (int rep = 0; rep < reps; ++rep) {
        if (rep == 1) t0 = clock64();  // rep 0 warms L2 / the instruction cache
        for (int layer = 0; layer < kLayers; ++layer) {
            load(f0, layer, 0);
            int kk = 0;
            if constexpr (INTER) {
                for (; kk + 2 < kKC; kk += 2) {
                    compute_interleaved(f0, f1, layer, kk + 1);
                    compute_interleaved(f1, f0, layer, kk + 2);
                }
                compute(f0);
                if (rep >= 1) mf += (kKC)*40;
                continue;
            }
            for (; kk + 2 < kKC; kk += 2) {
                load(f1, layer, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(f0);
                __builtin_amdgcn_sched_barrier(0);
                load(f0, layer, kk + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
            compute(f0);  // chunk 12
            if (rep >= 1) mf += (kKC)*40;
        }
    }
    asm volatile("s_nop 15" ::: "memory");
    for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(acc[i]));
    t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && lane == 0 && wave == 0) { cyc[0] = t1 - t0; cyc[1] = mf; }
}

template <int MASK>
void run(const char* name, const float* w, float* out, long long* cyc, long long wg_stride) {
    const int reps = 5, grid = 220;
    const size_t lds = kRows * kLd * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MASK>), dim3(grid), dim3(256), lds, 0, w, out, cyc, reps, wg_stride);  // warm
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((probe<MASK>), dim3(grid), dim3(256), lds, 0, w, out, cyc, reps, wg_stride);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    long long c[2];
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("{\"variant\": \"%s\", \"mask\": %d, \"cycles_per_mfma\": %.3f, \"launch_ms\": %.4f, \"mfmas_timed_per_wave\": %lld}\n", name, MASK,
           (double)c[0] / (double)c[1], ms, c[1]);
}

int main() {
    const long long member_floats = (long long)kLayers * 13 * kKC * 256;  // 8 "layers" x 13 column tiles x 13 chunks x 1 KiB
    const int grid = 220;
    float *w, *out;
    long long* cyc;
    hipMalloc(&w, (size_t)member_floats * 4 * grid);
    hipMalloc(&out, grid * 256 * 4);
    hipMalloc(&cyc, 64);
    std::vector<float> h((size_t)member_floats, 0.001f);
    for (int g = 0; g < grid; ++g) hipMemcpy(w + (size_t)g * member_floats, h.data(), (size_t)member_floats * 4, hipMemcpyHostToDevice);
    run<0>("MFMAs only", w, out, cyc, member_floats);
    run<1>("+ s_nop 1 per k-step", w, out, cyc, member_floats);
    run<2>("+ LDS fragment reads", w, out, cyc, member_floats);
    run<4>("+ weight fragment loads (shared block)", w, out, cyc, member_floats);
    run<6>("LDS + weight loads", w, out, cyc, member_floats);
    run<7>("LDS + weight loads + s_nop (the kernel's loop)", w, out, cyc, member_floats);
    run<15>("the kernel's loop, weights private per workgroup (L2 not shared)", w, out, cyc, member_floats);
    run<22>("LDS + weight loads, INTERLEAVED one per 5 MFMAs", w, out, cyc, member_floats);
    run<23>("interleaved + s_nop", w, out, cyc, member_floats);
    run<31>("interleaved + s_nop, weights private per workgroup", w, out, cyc, member_floats);
    return 0;
}
