"""Python restatement of the library's row-tile rule (mbrl-lib_amd/csrc/hipets.hip choose_R + wave_units, launch.hpp's instance tables) for
hid-200 models on a 256-CU chip -- test infrastructure: lets the calibration claim of DESIGN.md section 4 ("the fastest R in 23 of the 24
measured cases") be checked on CPU against the committed sweeps, and the restatement itself be checked against the library on the GPU
(tests/test_gpu_cost_model.py)."""
NUM_CU = 256
K_WAVES = 4


def wave_units(C, R):
    full, rem = divmod(C, K_WAVES)
    nu = rem * R
    simd = [0] * 4
    for w in range(K_WAVES):
        simd[w % 4] += full * R + ((nu - w + K_WAVES - 1) // K_WAVES if w < nu else 0)
    return max(simd)


def cost(tiles, slices, R, lean, desync, C=13, wide=False):
    a = ((6.45 if desync else 2.67) if wide else 1.77) * C / 13.0
    groups = (tiles + R - 1) // R
    nwg = groups * slices
    n = (nwg + NUM_CU - 1) // NUM_CU
    co = 2 if (R <= 2 and not wide) else 1
    u = wave_units(C, R)
    u_pair = C * R / 4.0 if (co == 2 and desync) else u
    full, rem = divmod(n, co)
    c = full * (a + co * u_pair) + ((a + rem * u) if rem else 0.0)
    return c * (1.0 if lean else 1.08)


def choose_r(pop, P, members, mode, lean_rs, rs=(1, 2, 3, 4), wide=False):
    """mode 'fast': the pop P rows in one run (since round 5; before: one slice of pop rows per particle -- the same choice for every
    workload below); 'device': one slice of pop P / members rows per member.  wide: the Humanoid-v4 instances (rs = (1, 2))."""
    fast = mode == "fast"
    tiles, slices = ((pop * P + 15) // 16, 1) if fast else ((pop * P // members + 15) // 16, members)
    best, best_cost = rs[0], float("inf")
    for R in rs:
        c = cost(tiles, slices, R, R in lean_rs, fast, wide=wide)
        if c < best_cost - 1e-9:
            best, best_cost = R, c
    return best


# (sweep file, key) -> pop, P, active members, row-tile counts with a fused instance per mode
WORKLOADS = {
    ("r4_device_r_sweep.json", "cfg5 (pop 2000 x 20, H 50)"): (2000, 20, 5, {1, 2, 3}),
    ("r4_device_r_sweep.json", "cfg2 x 2 (pop 1000 x 20, H 30)"): (1000, 20, 5, {1, 2, 3}),
    ("r4_device_r_sweep.json", "cfg2 x 4 (pop 2000 x 20, H 30)"): (2000, 20, 5, {1, 2, 3}),
    ("r4_device_r_sweep.json", "pets_halfcheetah x 2 (pop 800 x 20, H 30)"): (800, 20, 5, {1, 2, 3}),
    ("r4_device_r_sweep.json", "cfg4 first iCEM iteration (obs 45, pop 1036 x 20, H 40)"): (1036, 20, 5, {2, 3, 4}),
    ("r4_stock_workloads.json", "cfg2_synthetic"): (500, 20, 5, {1, 2, 3}),
    ("r4_stock_workloads.json", "stock_halfcheetah"): (400, 20, 5, {1, 2, 3}),
    ("r4_stock_workloads.json", "stock_cartpole"): (350, 20, 5, {1, 2}),
    ("r4_stock_workloads.json", "stock_pusher"): (350, 20, 5, {1, 2}),
    ("r4_stock_workloads.json", "stock_reacher"): (350, 20, 5, {1, 2}),
    ("r4_stock_workloads.json", "stock_mppi_halfcheetah_model"): (350, 20, 5, {1, 2}),
    ("r4_stock_workloads.json", "stock_inv_pendulum"): (480, 20, 5, {3}),
}
