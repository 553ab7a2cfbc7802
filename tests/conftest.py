import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _cache_synthetic_state_dicts():
    """`synth.make_state_dict` draws every tensor from its own generator (keyed by the tensor's position), so a tensor is the same
    whatever subset is asked for.  The GPU suite asks for the SD1.5 / SDXL / DiT weights a dozen times (698 M - 2.6 G parameters of
    host-side random numbers each): keep what has been drawn once per (config, seed) for the session and hand out dict views."""
    from diffsim_amd import synth as S
    orig = S.make_state_dict
    cache = {}

    def cached(cfg, seed=0, keys=None):
        store = cache.setdefault((repr(cfg), int(seed)), {})
        want = None if keys is None else list(keys)
        if want is None:
            full = orig(cfg, seed, None) if not store.get("__full__") else None
            if full is not None:
                store.update(full)
                store["__full__"] = True
            return {k: v for k, v in store.items() if k != "__full__"}
        missing = [k for k in want if k not in store]
        if missing:
            store.update(orig(cfg, seed, missing))
        return {k: store[k] for k in want if k in store}

    S.make_state_dict = cached
    yield
    S.make_state_dict = orig
