import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# the library's test / A-B hooks (RWARE_STAGGER_TICKS, RWARE_PIPE_GRID, RWARE_MULTI_THREADS, ...: csrc/rware_hooks.h) are honoured
# only with this switch on — tests reach paths through them that the measured rules would not take
os.environ["RWARE_HOOKS"] = "1"


import pytest  # noqa: E402


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`pytest tests -m "not gpu"` — the CPU suite, ~400 tests, most of them the product's kernels on host threads (every workgroup
    = 256 OS threads): ≈ 25 min on one core, ≈ 5 min on six.  When pytest-xdist is installed and the caller did not say otherwise
    (no -n, not --collect-only / --pdb, RWARE_TESTS_NO_AUTO_XDIST unset), the CPU suite spreads itself over the cores.  The GPU suite
    (`-m gpu`) is never touched: one device, one process."""
    opt = config.option
    if os.environ.get("RWARE_TESTS_NO_AUTO_XDIST") == "1" or not config.pluginmanager.hasplugin("xdist"):
        return None
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):   # (an xdist worker runs this hook too: never nest)
        return None
    if getattr(opt, "numprocesses", None) is not None or (getattr(opt, "markexpr", "") or "").replace(" ", "") != "notgpu":
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    opt.numprocesses = max(1, min(6, (os.cpu_count() or 2) - 1))   # (xdist's own hook, next in line, turns this into workers)
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver via gpurun)")


def _hip_device_visible() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) where no HIP device is visible — e.g. a plain `pytest tests` in the
    GPU-less build container.  On a GPU box nothing is skipped: a missing librware_hip.so then FAILS the tests."""
    import pytest

    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items or _hip_device_visible():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (the engine has no CPU fallback)")
    for it in gpu_items:
        it.add_marker(skip)
