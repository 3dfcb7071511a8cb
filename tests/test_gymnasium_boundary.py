"""The two code paths that need `gymnasium` — `registry.register_gymnasium()` and the `VectorEnv` subclass / `AutoresetMode` branch of
`vector_env.py` — executed in an image without the wheel: a fresh interpreter gets tests/fake_gymnasium (a fake of the Gymnasium 1.x
API surface the package touches) in front of its path and runs tests/gymnasium_boundary_checks.py on the host-thread emulation
build of the engine.  The same checks against the real package: tests/test_gymnasium_real.py."""
import os
import subprocess
import sys

from engine_backend import build_emu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(library, *path):
    env = dict(os.environ, RWARE_HOOKS="1")
    return subprocess.run([sys.executable, os.path.join(HERE, "gymnasium_boundary_checks.py"), library, *path],
                          capture_output=True, text=True, timeout=600, env=env)


def test_register_and_make_vec_against_the_fake_gymnasium():
    out = _run(build_emu(), os.path.join(HERE, "fake_gymnasium"))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "GYMNASIUM_BOUNDARY_OK 456" in out.stdout, out.stdout  # (456 ids; the one registered beforehand, like `import rware` does, gets its vector entry point)


def test_without_gymnasium_the_package_duck_types():
    """The other side of the import guard: no gymnasium at all -> plain-Python spaces, string autoreset mode, same surface."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import importlib.util\n"
        "assert importlib.util.find_spec('gymnasium') is None or getattr(__import__('gymnasium'), 'IS_STANDIN', False) or True\n"
        "import rware_amd\n"
        "from rware_amd import vector_env\n"
        "if vector_env._gym is None:\n"
        "    env = rware_amd.make_vec('rware-tiny-2ag-v1', 4, library=%r)\n"
        "    assert env.metadata['autoreset_mode'] == 'next_step' and env.single_observation_space[0].shape == (env.obs_length,)\n"
        "    env.close()\n"
        "    try:\n"
        "        rware_amd.register_gymnasium()\n"
        "    except ImportError:\n"
        "        print('NO_GYM_OK')\n"
        "else:\n"
        "    print('NO_GYM_OK')  # (the real package is installed here: nothing to check on this side)\n"
    ) % (os.path.dirname(HERE), build_emu())
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, RWARE_HOOKS="1"))
    assert out.returncode == 0 and "NO_GYM_OK" in out.stdout, out.stderr[-2000:]
