"""TEST-ONLY stand-in for `omegaconf` (attribute-access dict / list; no interpolation)."""


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __contains__(self, k):
        # omegaconf semantics: a key holding "???" is "missing" -> `k in cfg` is False
        return dict.__contains__(self, k) and not (
            isinstance(dict.__getitem__(self, k), str) and dict.__getitem__(self, k) == "???"
        )


class ListConfig(list):
    pass


def _wrap(x):
    if isinstance(x, dict):
        return DictConfig({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return ListConfig([_wrap(v) for v in x])
    return x


class OmegaConf:
    @staticmethod
    def create(x=None):
        return _wrap(x if x is not None else {})

    @staticmethod
    def to_container(x, **_k):
        return x
