"""ctypes binding of libhipets.so (include/hipets.h).  No CPU fallback: if the shared library
or a gfx950 device is missing, engine construction raises -- loudly, never silently."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPETS_LIB selects another build of the SAME library (kernel-variant experiments under profiles/); there is no fallback
LIB_PATH = os.environ.get("HIPETS_LIB") or os.path.join(_HERE, "libhipets.so")

ABI_VERSION = 6
MAX_LAYERS = 8

ACT = {"relu": 0, "silu": 1, "leaky_relu": 2, "tanh": 3, "sigmoid": 4}
PROP = {"random_model": 0, "fixed_model": 1, "expectation": 2}
OBS = {"none": 0, "halfcheetah": 1, "cartpole_pets": 2}
REW = {None: 0, "learned": 0, "cartpole": 1, "cartpole_pets": 2, "inverted_pendulum": 3, "halfcheetah": 4, "pusher": 5, "none": 6}
TERM = {"no_termination": 0, "cartpole": 1, "inverted_pendulum": 2, "hopper": 3, "walker2d": 4, "ant": 5, "humanoid": 6}
NORM = {"none": 0, "f32": 1, "f64": 2}
ENSEMBLE = {"gaussian_mlp": 0, "basic_ensemble": 1}
PREC = {"f32": 0, "bf16x3": 1}
MODE_EXACT, MODE_FAST, MODE_DEVICE = 0, 1, 2
MODES = {"exact": MODE_EXACT, "fast": MODE_FAST, "device": MODE_DEVICE}
KERNEL_CLASSES = ("generic", "hidden_static", "fused", "wide")  # HIPETS_KERNEL_*


class ModelDesc(C.Structure):
    _fields_ = [
        ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("in_dim", C.c_int32), ("out_dim", C.c_int32),
        ("hid", C.c_int32), ("n_layers", C.c_int32), ("ensemble_size", C.c_int32), ("n_members", C.c_int32),
        ("members", C.POINTER(C.c_int32)),
        ("activation", C.c_int32), ("leaky_slope", C.c_float), ("propagation", C.c_int32),
        ("deterministic", C.c_int32), ("obs_process", C.c_int32), ("reward_fn", C.c_int32),
        ("termination_fn", C.c_int32), ("target_is_delta", C.c_int32), ("learned_rewards", C.c_int32),
        ("n_no_delta", C.c_int32), ("no_delta", C.POINTER(C.c_int32)),
        ("normalizer", C.c_int32), ("norm_mean", C.POINTER(C.c_double)), ("norm_std", C.POINTER(C.c_double)),
        ("min_logvar", C.POINTER(C.c_float)), ("max_logvar", C.POINTER(C.c_float)),
        ("weights", C.POINTER(C.c_void_p)), ("biases", C.POINTER(C.c_void_p)),
        ("ensemble_kind", C.c_int32),
        ("precision", C.c_int32),
    ]


class RolloutOpts(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("perms", C.c_void_p), ("eps", C.c_void_p),
        ("seed", C.c_uint64), ("stream_id", C.c_uint64),
        ("member_schedule", C.c_void_p), ("fast_eps", C.c_void_p),
        ("trace_next_obs", C.c_void_p), ("trace_rewards", C.c_void_p),
        ("rows_per_group", C.c_int32),
        ("phase_cycles", C.c_void_p),
        ("no_sample", C.c_int32),
        ("rows_per_member", C.c_int32),
        ("n_env", C.c_int32),
        ("generic_kernel", C.c_int32),
        ("member_schedule_len", C.c_int32),
        ("perm_stream_id", C.c_uint64),
    ]


class CemParams(C.Structure):
    _fields_ = [
        ("population_size", C.c_int32), ("horizon", C.c_int32), ("act_dim", C.c_int32),
        ("num_iterations", C.c_int32), ("elite_num", C.c_int32), ("alpha", C.c_double),
        ("return_mean_elites", C.c_int32), ("clipped_normal", C.c_int32), ("unbiased_var", C.c_int32),
    ]


class IcemParams(C.Structure):
    _fields_ = [
        ("population_size", C.c_int32), ("horizon", C.c_int32), ("act_dim", C.c_int32),
        ("num_iterations", C.c_int32), ("elite_num", C.c_int32), ("keep_elite_size", C.c_int32),
        ("population_size_module", C.c_int32), ("return_mean_elites", C.c_int32), ("alpha", C.c_double),
        ("population_decay_factor", C.c_double), ("colored_noise_exponent", C.c_double),
    ]


class PlanTrace(C.Structure):
    _fields_ = [("populations", C.c_void_p), ("values", C.c_void_p), ("mus", C.c_void_p), ("dispersions", C.c_void_p),
                ("elite_idx", C.c_void_p), ("max_rows", C.c_int32)]


_PLANET_TENSORS = ("w_embed", "b_embed", "w_ih", "b_ih", "w_hh", "b_hh", "w_prior1", "b_prior1", "w_prior2", "b_prior2",
                   "w_rew1", "b_rew1", "w_rew2", "b_rew2", "w_rew3", "b_rew3")


class PlanetDesc(C.Structure):
    _fields_ = [("latent_size", C.c_int32), ("action_size", C.c_int32), ("belief_size", C.c_int32), ("hidden_size", C.c_int32),
                ("min_std", C.c_float)] + [(n, C.c_void_p) for n in _PLANET_TENSORS]


class PlanetOpts(C.Structure):
    _fields_ = [("eps", C.c_void_p), ("seed", C.c_uint64), ("stream_id", C.c_uint64), ("no_sample", C.c_int32),
                ("trace_latent", C.c_void_p), ("trace_belief", C.c_void_p), ("trace_rewards", C.c_void_p), ("phase_cycles", C.c_void_p)]


# every symbol include/hipets.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "hipets_abi_version": (C.c_int, []),
    "hipets_last_error": (C.c_char_p, []),
    "hipets_last_error_kind": (C.c_int, []),
    "hipets_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "hipets_destroy": (None, [_P]),
    "hipets_set_model": (C.c_int, [_P, C.POINTER(ModelDesc), _P]),
    "hipets_rollout": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(RolloutOpts), _P, _P]),
    "hipets_step": (C.c_int, [_P, _P, _P, C.c_int32, C.POINTER(RolloutOpts), _P, _P, _P, _P]),
    "hipets_fast_geometry": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hipets_kernel_class": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hipets_fast_schedule": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_fast_normals": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_device_perms": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_set_plan_mode": (C.c_int, [_P, C.c_int32]),
    "hipets_set_persistent": (C.c_int, [_P, C.c_int32]),
    "hipets_set_handover_timeout": (C.c_int, [_P, C.c_double]),
    "hipets_check_async_error": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "hipets_set_plan_trace": (C.c_int, [_P, C.POINTER(PlanTrace)]),
    "hipets_cem_sample": (C.c_int, [_P, C.POINTER(CemParams), _P, _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_cem_refit": (C.c_int, [_P, C.POINTER(CemParams), _P, _P, _P, _P, _P, _P, _P, _P]),
    "hipets_cem_refit_elites": (C.c_int, [_P, C.POINTER(CemParams), _P, _P, _P, _P, _P, _P, _P, _P]),
    "hipets_gather_rows": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "hipets_mppi_sample": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_mppi_update": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, _P]),
    "hipets_icem_sample": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_icem_shift": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_plan_cem": (C.c_int, [_P, C.POINTER(CemParams), _P, _P, _P, _P, C.c_int32, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_plan_cem_batched": (C.c_int, [_P, C.POINTER(CemParams), C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_plan_mppi": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, _P, _P, _P, _P, C.c_int32,
                                   C.c_uint64, C.c_uint64, _P]),
    "hipets_plan_mppi_batched": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, _P, _P, _P, _P,
                                           C.c_int32, C.c_uint64, C.c_uint64, _P]),
    "hipets_plan_icem_batched": (C.c_int, [_P, C.POINTER(IcemParams), C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P, C.c_int32, C.c_uint64,
                                           C.c_uint64, _P, _P]),
    "hipets_plan_icem": (C.c_int, [_P, C.POINTER(IcemParams), _P, _P, _P, _P, C.c_int32, _P, _P, C.c_int32, C.c_uint64, C.c_uint64,
                                   _P, _P]),
    "hipets_comm_unique_id": (C.c_int, [_P]),
    "hipets_comm_init": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "hipets_comm_destroy": (C.c_int, [_P]),
    "hipets_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hipets_plan_cem_sharded": (C.c_int, [_P, C.POINTER(CemParams), _P, _P, _P, _P, C.c_int32, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_plan_mppi_sharded": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, _P, _P, _P, _P, C.c_int32,
                                           C.c_uint64, C.c_uint64, _P]),
    "hipets_plan_icem_sharded": (C.c_int, [_P, C.POINTER(IcemParams), _P, _P, _P, _P, C.c_int32, _P, _P, C.c_int32, C.c_uint64, C.c_uint64,
                                           _P, _P]),
    "hipets_planet_set_model": (C.c_int, [_P, C.POINTER(PlanetDesc), _P]),
    "hipets_planet_rollout": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(PlanetOpts), _P, _P]),
    "hipets_plan_planet_cem": (C.c_int, [_P, C.POINTER(CemParams), _P, _P, _P, _P, _P, C.c_int32, C.c_uint64, C.c_uint64, _P, _P]),
    "hipets_timing_enable": (C.c_int, [_P, C.c_int32]),
    "hipets_timing_read": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int32]),
}

_lib = None


COMM_ID_BYTES = 128


ERR_NONE, ERR_INVALID_ARGUMENT, ERR_RUNTIME, ERR_TIMEOUT = 0, 1, 2, 3


class HipetsError(RuntimeError):
    """A failed libhipets call.  ``kind`` = hipets_last_error_kind(): ERR_INVALID_ARGUMENT (the library refused the arguments: the
    same on every rank that passed them), ERR_RUNTIME (a HIP / RCCL call, an allocation or a launch failed), ERR_TIMEOUT.  An error
    raised on the Python side without a kind (a missing library, an engine without a model) counts as a runtime failure: only the C
    library's own verdict marks an error as a deterministic rejection of the arguments (hipets.dist.run_sharded relies on that)."""

    def __init__(self, message="", kind: int = ERR_RUNTIME):
        super().__init__(message)
        self.kind = kind


def load():
    """dlopen libhipets.so and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipetsError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "hipets has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.hipets_abi_version() != ABI_VERSION:
        raise HipetsError(f"libhipets ABI {lib.hipets_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        lib = load()
        raise HipetsError(lib.hipets_last_error().decode(), lib.hipets_last_error_kind())
