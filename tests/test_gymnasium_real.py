"""tests/gymnasium_boundary_checks.py against the REAL gymnasium, wherever the wheel exists (it does not in the build image: skipped
there; tests/test_gymnasium_boundary.py runs the same checks on a fake of the API).  On a GPU box the checks construct the envs on the
gfx950 library, elsewhere on the host-thread emulation build."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_register_and_make_vec_against_real_gymnasium():
    gym = pytest.importorskip("gymnasium")
    if getattr(gym, "IS_STANDIN", False):
        pytest.skip("only the oracle's stand-in is importable")
    if not hasattr(gym, "make_vec"):
        pytest.skip(f"gymnasium {gym.__version__} has no make_vec (needs >= 1.0)")
    try:
        import torch

        on_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        on_gpu = False
    if on_gpu:
        library = "-"
    else:
        from engine_backend import build_emu

        library = build_emu()
    out = subprocess.run([sys.executable, os.path.join(HERE, "gymnasium_boundary_checks.py"), library], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, RWARE_HOOKS="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "GYMNASIUM_BOUNDARY_OK" in out.stdout
