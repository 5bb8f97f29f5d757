// What does ONE exchange of an activation slice between two workgroups cost on gfx950 -- the primitive a 2-CU column split of a
// one-tile step would pay once per dependent GEMM (round-5 verdict, item 4b: PlaNet pop 1000 runs 63 one-tile workgroups at 38.8 us
// per step against a one-CU MFMA floor of ~20 us; two CUs per row tile would halve the MFMA time and add one exchange per op).
//
// Two workgroups (256 threads each) ping-pong: A publishes N 16-byte {value, tag, value, tag} pairs (what the rollout kernel's
// hand-over uses: tagged granules, write-through stores, polled loads, no fence), B waits until all N carry the round's tag, publishes
// its own N pairs with the same tag, A waits for those, next round.  One round = TWO one-way exchanges.  Reported per one-way exchange,
// for the pair of workgroups on the SAME XCD (blocks 0 and 8: hardware deals block b to XCD b % 8) and on DIFFERENT XCDs (blocks 0 and
// 1), for slices of 16 B, 1.6 KB (16 rows x 25 fp32: a quarter of a 200-wide hidden layer per wave), 6.4 KB (16 x 100: half a hidden
// layer -- what each CU of a 2-way column split sends per op) and 12.8 KB (16 x 200), with the store / load scopes:
//   sc1/sc1   device scope both ways (what the rollout kernel uses: correct for any placement)
//   (a plain store + sc0 / workgroup-scope load variant was tried for the same-XCD pair in round 6: the load may be served from the
//   reader's L1 and never see the partner's store -- the run spun until its time-out.  Device scope is the cheapest correct one.)
//   hipcc --offload-arch=gfx950 -O3 pair_exchange.hip -o pair_exchange && ./pair_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int SCOPE>  // 0: sc1 store / sc1 load; 1: plain store / sc0 load
__device__ __forceinline__ void st_pair(u32x4* p, u32x4 v) {
    if (SCOPE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int SCOPE>
__device__ __forceinline__ u32x4 ld_pair(const u32x4* p) {
    u32x4 v;
    if (SCOPE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// block `a` and block `b` exchange; every other block exits.  buf: [2][n_pairs] pairs (slot 0: written by a, slot 1: by b)
template <int SCOPE>
__global__ __launch_bounds__(256) void k(u32x4* buf, int n_pairs, int rounds, int a, int b, long long* ticks, int* bad) {
    const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    u32x4* mine = buf + (size_t)me * n_pairs;
    const u32x4* theirs = buf + (size_t)(1 - me) * n_pairs;
    const int tid = threadIdx.x;
    long long t0 = 0;
    for (int r = 1; r <= rounds + 8; ++r) {
        if (r == 9) { __syncthreads(); t0 = wall_clock64(); }  // 8 warm-up rounds
        const unsigned tag = (unsigned)r;
        if (me == 0)  // a publishes first
            for (int i = tid; i < n_pairs; i += 256) st_pair<SCOPE>(mine + i, u32x4{(unsigned)i, tag, (unsigned)i + 1u, tag});
        // wait for the partner's pairs of this round
        for (int i = tid; i < n_pairs; i += 256) {
            long long spins = 0;
            for (;;) {
                const u32x4 v = ld_pair<SCOPE>(theirs + i);
                if (v[1] == tag && v[3] == tag) { if (v[0] != (unsigned)i) *bad = 1; break; }
                if (++spins > 200000) { *bad = 2; break; }
            }
        }
        __syncthreads();  // the whole slice has arrived (the consumer of an activation slice needs all of it)
        if (me == 1)
            for (int i = tid; i < n_pairs; i += 256) st_pair<SCOPE>(mine + i, u32x4{(unsigned)i, tag, (unsigned)i + 1u, tag});
    }
    __syncthreads();
    if (tid == 0) ticks[me] = wall_clock64() - t0;  // 100 MHz
}

template <int SCOPE>
double run(int n_pairs, int a, int b) {
    u32x4* buf; long long* ticks; int* bad;
    hipMalloc(&buf, (size_t)2 * n_pairs * 16); hipMemset(buf, 0, (size_t)2 * n_pairs * 16);
    hipMalloc(&ticks, 16); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    const int rounds = 500;
    hipLaunchKernelGGL(k<SCOPE>, dim3(16), dim3(256), 0, 0, buf, n_pairs, rounds, a, b, ticks, bad);
    hipDeviceSynchronize();
    long long t[2]; int hb = 0;
    hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    hipFree(buf); hipFree(ticks); hipFree(bad);
    if (hb) return -hb;
    return (double)t[0] * 10.0 / (2.0 * rounds);  // ns per ONE-WAY exchange (10 ns per tick, two exchanges per round)
}

int main() {
    printf("{\"what\": \"one-way exchange of a tagged-pair slice between two workgroups, ns (wall clock, 500 rounds)\"");
    const int sizes[] = {1, 100, 400, 800};  // pairs: 16 B, 1.6 KB, 6.4 KB, 12.8 KB
    for (int s = 0; s < 4; ++s) {
        const int n = sizes[s];
        printf(", \"%d_bytes\": {\"same_xcd_sc1\": %.0f, \"other_xcd_sc1\": %.0f}", n * 16, run<0>(n, 0, 8), run<0>(n, 0, 1));
        fflush(stdout);
    }
    printf("}\n");
    return 0;
}
