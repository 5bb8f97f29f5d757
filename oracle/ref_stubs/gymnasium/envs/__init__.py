from . import mujoco  # noqa: F401
