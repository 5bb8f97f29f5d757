class EzPickle:
    def __init__(self, *a, **k):
        pass


class seeding:  # noqa: N801
    @staticmethod
    def np_random(seed=None):
        import numpy as np

        return np.random.default_rng(seed), seed
