"""hipets -- MI355X-native PETS planning / rollout engine behind mbrl-lib's own plugin seams.

Public surface (names follow mbrl-lib so the stock Hydra configs only swap ``_target_``):
``Engine``, ``ModelSpec``, ``spec_from_model_env``, and (planning.py) ``CEMOptimizer``,
``TrajectoryOptimizer``, ``TrajectoryOptimizerAgent``, ``make_eval_fn``,
``create_trajectory_optim_agent_for_model``.
"""
from ._lib import HipetsError, LIB_PATH  # noqa: F401
from .model import ModelSpec, UnsupportedModelError, model_version, spec_from_model_env  # noqa: F401
from .engine import Engine  # noqa: F401

__version__ = "0.1.0"
