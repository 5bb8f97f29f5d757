import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mbrl-lib_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def to_spec(om, obs_dim, act_dim, **kw):
    """OracleModel (test infra) -> hipets.ModelSpec (product).  The product never sees the oracle."""
    import hipets

    return hipets.ModelSpec(
        weights=om.weights, biases=om.biases, obs_dim=obs_dim, act_dim=act_dim, min_logvar=om.min_logvar,
        max_logvar=om.max_logvar, elite_models=om.elite_models, activation=om.activation, propagation=om.propagation,
        deterministic=om.deterministic, norm_mean=om.norm_mean, norm_std=om.norm_std, target_is_delta=om.target_is_delta,
        no_delta_list=om.no_delta_list, learned_rewards=om.learned_rewards, obs_process=om.obs_process, reward=om.reward,
        termination=om.termination, ensemble_kind=om.ensemble_kind, **kw,
    )


@pytest.fixture(scope="session")
def engine():
    import torch

    import hipets

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return hipets.get_engine("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    """Write the oracle memo's new entries (tests/oracle_cache.py) where HIPETS_ORACLE_CACHE_OUT says, and report its hit rate."""
    try:
        import oracle_cache

        oracle_cache.flush_all()
        if oracle_cache.stats["hits"] or oracle_cache.stats["misses"]:
            print(f"\n[oracle_cache] hits {oracle_cache.stats['hits']}, misses {oracle_cache.stats['misses']}, "
                  f"recomputed and compared with the stored entry {oracle_cache.stats.get('verified', 0)}")
    except Exception as exc:  # never fail a run over the memo
        print(f"[oracle_cache] {exc}")
