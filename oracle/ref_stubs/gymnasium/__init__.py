"""TEST-ONLY stand-in for `gymnasium` (only the names the reference touches at import time)."""
import numpy as np

from . import spaces, wrappers, error, logger, envs, utils  # noqa: F401


class Env:
    observation_space = None
    action_space = None
    metadata = {}

    def reset(self, *a, **k):
        raise NotImplementedError

    def step(self, *a, **k):
        raise NotImplementedError


class Wrapper(Env):
    def __init__(self, env):
        self.env = env


def make(*_a, **_k):
    raise error.DependencyNotInstalled("gymnasium stand-in: no real environments")
