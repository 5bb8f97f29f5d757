"""TEST-ONLY stand-in for `hydra` so the unmodified reference imports in this container.
Object construction only -- contributes no arithmetic."""
from . import utils  # noqa: F401


def main(*_a, **_k):
    def deco(fn):
        return fn

    return deco
