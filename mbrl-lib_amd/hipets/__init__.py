"""hipets -- MI355X-native PETS planning / rollout engine behind mbrl-lib's own plugin seams.

Public names follow mbrl-lib so the stock Hydra configs only swap ``_target_``:
``hipets.TrajectoryOptimizerAgent``, ``hipets.CEMOptimizer`` ... or, on a stock agent,
``agent.set_trajectory_eval_fn(hipets.make_eval_fn(model_env, num_particles))``.
"""
from ._lib import ERR_INVALID_ARGUMENT, ERR_NONE, ERR_RUNTIME, ERR_TIMEOUT, HipetsError, LIB_PATH  # noqa: F401
from .model import (  # noqa: F401
    ModelSpec,
    PlaNetSpec,
    UnsupportedModelError,
    model_version,
    spec_from_checkpoint,
    spec_from_model_env,
    spec_from_planet_model,
)
from .engine import Engine  # noqa: F401
from .planning import (  # noqa: F401
    Agent,
    BatchedCEMAgent,
    BatchedICEMAgent,
    BatchedMPPIAgent,
    CEMOptimizer,
    HipTrajectoryEvalFn,
    ICEMOptimizer,
    MPPIOptimizer,
    ModelEnv,
    Optimizer,
    PlaNetTrajectoryEvalFn,
    TrajectoryOptimizer,
    TrajectoryOptimizerAgent,
    UnfusedTrajectoryEvalFn,
    complete_agent_cfg,
    create_trajectory_optim_agent_for_model,
    get_engine,
    make_eval_fn,
)
from .propagation import (  # noqa: F401
    propagate,
    propagate_expectation,
    propagate_fixed_model,
    propagate_from_indices,
    propagate_random_model,
)
from . import dist  # noqa: F401

__version__ = "0.1.0"
