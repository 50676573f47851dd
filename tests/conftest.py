import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "prompt-cache_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (GPU tests run via gpurun / the driver)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _oracle_blas_width():
    """The numpy oracle's BLAS pool at the width it runs fastest at on a many-core host: OpenBLAS defaults to 64 threads on the
    256-CPU GPU boxes, where an sgemm of a few hundred rows is 4-5x slower than on 16 (tools/blas_probe.py: 0.54 vs 2.6-2.8
    TFLOP/s at 130 rows; the seven full-depth parity tests took 490 s of the suite's 656 s that way)."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:  # pragma: no cover
        yield
        return
    if (os.cpu_count() or 1) > 32:
        with threadpool_limits(limits=16, user_api="blas"):
            yield
    else:
        yield
