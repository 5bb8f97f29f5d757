class MujocoEnv:
    pass


class mujoco_env:  # noqa: N801
    MujocoEnv = MujocoEnv
