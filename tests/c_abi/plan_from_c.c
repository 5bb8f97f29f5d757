/* The drop-in boundary is plain C: this program uses include/hipets.h and the HIP runtime's C API only (no C++, no
 * torch).  It builds a zero-weight deterministic ensemble, for which the rollout has a closed form,
 *     next_obs = obs (delta targets, zero prediction),  reward = obs[0] - 0.1 |a|^2   (mbrl/env/reward_fns.py:33-38)
 * so  return(candidate) = sum_t (s0[0] - 0.1 |a_t|^2)  for every particle, and checks hipets_rollout (EXACT and FAST)
 * against it, then runs one fused hipets_plan_cem whose optimum is the zero plan.
 * Built by __graft_entry__.build() with gcc; run by tests/test_gpu_c_abi.py on the GPU box. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "hipets.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_LIB(x) do { if ((x) != 0) { fprintf(stderr, "hipets error at %s:%d: %s\n", __FILE__, __LINE__, hipets_last_error()); return 3; } } while (0)

enum { OBS = 5, ACT = 2, HID = 24, E = 3, LAYERS = 3, POP = 37, P = 3, H = 6 };

static float lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFF) / 65535.0f * 2.0f - 1.0f; }

int main(void) {
    if (hipets_abi_version() != HIPETS_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    hipets_engine* eng = NULL;
    CHECK_LIB(hipets_create(0, &eng));

    /* zero weights / biases: [E, in_l, out_l] and [E, 1, out_l] */
    const int in_dims[LAYERS] = {OBS + ACT, HID, HID};
    const int out_dims[LAYERS] = {HID, HID, OBS}; /* deterministic head: out = obs */
    void* w[LAYERS];
    void* b[LAYERS];
    for (int l = 0; l < LAYERS; ++l) {
        CHECK_HIP(hipMalloc(&w[l], sizeof(float) * E * in_dims[l] * out_dims[l]));
        CHECK_HIP(hipMemset(w[l], 0, sizeof(float) * E * in_dims[l] * out_dims[l]));
        CHECK_HIP(hipMalloc(&b[l], sizeof(float) * E * out_dims[l]));
        CHECK_HIP(hipMemset(b[l], 0, sizeof(float) * E * out_dims[l]));
    }
    int32_t members[E] = {0, 1, 2};
    hipets_model_desc d;
    memset(&d, 0, sizeof d);
    d.obs_dim = OBS; d.act_dim = ACT; d.in_dim = OBS + ACT; d.out_dim = OBS; d.hid = HID; d.n_layers = LAYERS;
    d.ensemble_size = E; d.n_members = E; d.members = members;
    d.activation = HIPETS_ACT_SILU; d.propagation = HIPETS_PROP_RANDOM_MODEL; d.deterministic = 1;
    d.obs_process = HIPETS_OBS_NONE; d.reward_fn = HIPETS_REW_HALFCHEETAH; d.termination_fn = HIPETS_TERM_NONE;
    d.target_is_delta = 1; d.normalizer = HIPETS_NORM_NONE;
    d.weights = (const void* const*)w; d.biases = (const void* const*)b;
    d.ensemble_kind = HIPETS_ENSEMBLE_GAUSSIAN_MLP;
    CHECK_LIB(hipets_set_model(eng, &d, NULL));

    /* random action sequences, known-answer returns */
    static float actions[POP * H * ACT];
    static float expect[POP];
    float s0[OBS] = {0.7f, -0.2f, 0.1f, 0.3f, -0.4f};
    uint32_t seed = 12345u;
    for (int c = 0; c < POP; ++c) {
        float ret = 0.f;
        for (int t = 0; t < H; ++t) {
            float sq = 0.f;
            for (int a = 0; a < ACT; ++a) { float v = lcg(&seed); actions[(c * H + t) * ACT + a] = v; sq += v * v; }
            ret += s0[0] + (-0.1f * sq);
        }
        expect[c] = ret;
    }
    float *d_actions, *d_returns;
    CHECK_HIP(hipMalloc((void**)&d_actions, sizeof actions));
    CHECK_HIP(hipMemcpy(d_actions, actions, sizeof actions, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc((void**)&d_returns, sizeof(float) * POP));

    /* B = POP * P = 111 rows = 37 per member: EXACT mode wants a permutation per step */
    static int64_t perms[H * POP * P];
    for (int t = 0; t < H; ++t)
        for (int j = 0; j < POP * P; ++j) perms[t * POP * P + j] = (j * 7 + t) % (POP * P); /* 7 is coprime with 111 */
    int64_t* d_perms;
    CHECK_HIP(hipMalloc((void**)&d_perms, sizeof perms));
    CHECK_HIP(hipMemcpy(d_perms, perms, sizeof perms, hipMemcpyHostToDevice));

    float got[POP];
    const int modes[2] = {HIPETS_MODE_EXACT, HIPETS_MODE_FAST};
    for (int m = 0; m < 2; ++m) {
        hipets_rollout_opts o;
        memset(&o, 0, sizeof o);
        o.mode = modes[m];
        o.perms = modes[m] == HIPETS_MODE_EXACT ? d_perms : NULL;
        o.seed = 9; o.stream_id = 1;
        CHECK_LIB(hipets_rollout(eng, d_actions, s0, POP, H, P, &o, d_returns, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipMemcpy(got, d_returns, sizeof got, hipMemcpyDeviceToHost));
        for (int c = 0; c < POP; ++c)
            if (!(fabsf(got[c] - expect[c]) <= 1e-5f * fmaxf(1.f, fabsf(expect[c])))) {
                fprintf(stderr, "mode %d candidate %d: got %.7f expected %.7f\n", modes[m], c, got[c], expect[c]);
                return 4;
            }
    }

    /* one fused CEM plan: the objective sum_t (s0[0] - 0.1 |a_t|^2) is maximised by the zero plan */
    hipets_cem_params cp;
    memset(&cp, 0, sizeof cp);
    cp.population_size = 300; cp.horizon = H; cp.act_dim = ACT; cp.num_iterations = 10; cp.elite_num = 30; cp.alpha = 0.1;
    cp.return_mean_elites = 1; cp.clipped_normal = 0; cp.unbiased_var = 1;
    float lower[H * ACT], upper[H * ACT], x0[H * ACT], plan[H * ACT];
    for (int i = 0; i < H * ACT; ++i) { lower[i] = -1.f; upper[i] = 1.f; x0[i] = 0.5f; }
    float *d_lower, *d_upper, *d_x0, *d_plan;
    CHECK_HIP(hipMalloc((void**)&d_lower, sizeof lower)); CHECK_HIP(hipMemcpy(d_lower, lower, sizeof lower, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc((void**)&d_upper, sizeof upper)); CHECK_HIP(hipMemcpy(d_upper, upper, sizeof upper, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc((void**)&d_x0, sizeof x0)); CHECK_HIP(hipMemcpy(d_x0, x0, sizeof x0, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc((void**)&d_plan, sizeof plan));
    CHECK_LIB(hipets_plan_cem(eng, &cp, d_x0, d_lower, d_upper, s0, P, 3, 1, d_plan, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(plan, d_plan, sizeof plan, hipMemcpyDeviceToHost));
    float worst = 0.f;
    for (int i = 0; i < H * ACT; ++i) worst = fmaxf(worst, fabsf(plan[i]));
    if (!(worst < 0.2f)) { fprintf(stderr, "CEM did not approach the zero plan: max |a| = %f\n", worst); return 5; }

    hipets_destroy(eng);
    printf("c_abi ok: rollout known-answer exact+fast, fused CEM plan max|a| = %.4f\n", worst);
    return 0;
}
