import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """Build (if stale) and load the HIP library; CPU-safe (hipcc cross-compiles, dlopen needs no GPU)."""
    from tacotron2_amd import build, native
    build.build(verbose=False)
    return native.load()


# The engine's module-level switches select launch forms for the WHOLE process.  A test that flips one and restores it wrongly
# silently changes what every later in-process test runs (test_zz6 restored TRAIN_BWD_PERSISTENT from an
# environment default that disagreed with engine.py's, so everything collected after it ran the opt-in backward form).  Every
# test therefore ends with the flags it started the SESSION with -- checked here, after each test, for all of them.
_ENGINE_FLAGS = ("TRAIN_FWD_PERSISTENT", "ENCODER_BATCH_PERSISTENT", "ENCODER_BATCH_PERSISTENT_TRAIN",
                 "ENCODER_BWD_PERSISTENT", "WGRAD16", "WGRAD_KK", "CONV16", "FAST_GRAD_GEMM", "ARENA", "WEIGHT_GUARD",
                 "COMPACT_BATCH", "PERSISTENT_DECODE", "PERSISTENT_ENCODER", "SMALL_BATCH_PERSISTENT", "DGRAD_SPLIT",
                 "ENC_DGRAD_SPLIT", "TRAIN_FWD_REPROMOTE_AFTER", "BN_BWD_IMAGE", "BN_FWD_IMAGE", "BIAS_GRAD16", "GATE_GRADS_BF16_ONLY", "DXD_RING")
_engine_flags_at_import = {}


@pytest.fixture(autouse=True)
def _engine_flags_are_restored():
    eng = sys.modules.get("tacotron2_amd.engine")
    if eng is not None and not _engine_flags_at_import:
        _engine_flags_at_import.update({k: getattr(eng, k) for k in _ENGINE_FLAGS if hasattr(eng, k)})
    yield
    eng = sys.modules.get("tacotron2_amd.engine")
    if eng is None:
        return
    if not _engine_flags_at_import:          # imported by this very test: its values at the end of it are the baseline
        _engine_flags_at_import.update({k: getattr(eng, k) for k in _ENGINE_FLAGS if hasattr(eng, k)})
        return
    changed = {k: (v, getattr(eng, k)) for k, v in _engine_flags_at_import.items() if getattr(eng, k) != v}
    for k, (v, _) in changed.items():        # do not let one offender fail every later test as well
        setattr(eng, k, v)
    dem = getattr(eng, "_DEMOTION", None)
    if dem is not None and dem.get("active"):  # a demotion left behind would re-promote (change forms) in the middle of a later test
        dem.update(active=False, saved=None, clean=0)
    assert not changed, "engine flags left changed by this test (import-time value, value left behind): %r" % (changed,)
