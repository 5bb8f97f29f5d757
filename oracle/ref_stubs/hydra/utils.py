import importlib


def _locate(path: str):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


get_method = _locate
get_class = _locate


def instantiate(cfg, *args, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg = {k: v for k, v in cfg.items() if not (isinstance(v, str) and v == "???")}
    cfg.update(kwargs)
    return _locate(target)(*args, **cfg)
