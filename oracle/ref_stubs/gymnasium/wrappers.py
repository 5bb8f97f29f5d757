class TimeLimit:
    def __init__(self, env, max_episode_steps=None):
        self.env = env
        self._max_episode_steps = max_episode_steps
