"""The C-ABI library loads and exports every entry point include/rware_hip.h declares; the pure
host entry points behave; without a HIP device construction fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rware_amd
from rware_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rware_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rw_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("rw_create", "rw_destroy", "rw_reset", "rw_step", "rw_step_device", "rw_step_many_device",
                 "rw_sync", "rw_get_buffer", "rw_read", "rw_write", "rw_recalc_grid", "rw_last_error"):
        assert must in names
    assert sorted(_capi.EXPORTS) == names, "python binding and header disagree"


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} missing from {_capi.DEFAULT_LIBRARY}"
    assert lib.rw_abi_version() == _capi.RW_ABI_VERSION


@pytest.mark.parametrize("seed", [0, 1, 7, 123456789, 2**32 - 1, 2**32 + 5, 2**63])
def test_rw_seed_state_matches_numpy(seed):
    st = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed))).bit_generator.state
    s, inc, m = st["state"]["state"], st["state"]["inc"], (1 << 64) - 1
    want = np.array([s >> 64, s & m, inc >> 64, inc & m, 0, 0], dtype=np.uint64)
    assert np.array_equal(_capi.seed_state(seed), want)


def test_bad_config_is_rejected_before_touching_the_device():
    lib = _capi.load()
    cfg = _capi.RwConfig()
    h = C.c_void_p()
    assert lib.rw_create(C.byref(cfg), C.byref(h)) == _capi.RW_ERR_INVALID_ARG  # abi_version 0
    assert b"abi_version" in lib.rw_last_error(None)
    assert lib.rw_create(None, C.byref(h)) == _capi.RW_ERR_INVALID_ARG
    assert lib.rw_step(None, None) == _capi.RW_ERR_INVALID_ARG
    assert lib.rw_destroy(None) == _capi.RW_OK


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="a HIP device is visible")
def test_no_device_means_loud_failure_not_a_cpu_path():
    with pytest.raises(_capi.EngineError) as ei:
        rware_amd.make_vec("rware-tiny-2ag-v1", 4)
    assert ei.value.code == _capi.RW_ERR_NO_DEVICE


def test_missing_library_is_a_loud_error(tmp_path):
    with pytest.raises(RuntimeError, match="not found"):
        _capi.load(str(tmp_path / "librware_hip.so"))


def test_kernel_isa_has_no_operand_order_sensitive_dpp_folds(tmp_path):
    """hipcc folds a DPP cross-lane move into the instruction that uses it.  For a NON-commutative use the fold has to
    pick the reversed opcode, and `v_subrev_u32_dpp` came out with its operands swapped on gfx950 (round 2: a winner test
    written as `x_k - x_me - 1` passed the host emulation and failed a golden trace on the GPU;
    profiles/tools/dpp_subrev_probe.hip shows it in isolation).  The exact-shape kernels may only contain DPP forms whose
    operand order cannot matter."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    csrc = os.path.join(ROOT, "robotic-warehouse_amd", "csrc")
    # the exact-shape kernels live in rware_static.hip, one translation unit per table group: all groups, side by side
    n_groups = int(re.search(r"kStaticGroups = (\d+)", open(os.path.join(csrc, "rware_static_table.h")).read()).group(1))
    procs = []
    for g in range(n_groups):
        procs.append(subprocess.Popen(
            [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + csrc, "-mllvm", "-amdgpu-kernarg-preload-count=16",
             f"-DRW_STATIC_GROUP={g}", "--cuda-device-only", "-S", "-o", str(tmp_path / f"static_g{g}.s"),
             os.path.join(csrc, "rware_static.hip")], stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for g, pr in enumerate(procs):
        _, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, err.decode()[-2000:]
    listing = "".join((tmp_path / f"static_g{g}.s").read_text() for g in range(n_groups))

    class _Out:  # (the checks below read one listing)
        @staticmethod
        def read_text():
            return listing
    out = _Out
    ops = set(re.findall(r"^\s*(v_[a-z0-9_]+_dpp)\b", out.read_text(), flags=re.M))
    assert ops, "the exact-shape builds exchange through DPP moves: none found?"
    allowed = {"v_mov_b32_dpp", "v_or_b32_dpp", "v_and_b32_dpp", "v_xor_b32_dpp", "v_add_u32_dpp", "v_max_i32_dpp", "v_max_u32_dpp",
               "v_min_i32_dpp", "v_min_u32_dpp"}
    assert ops <= allowed, f"operand-order-sensitive DPP folds in the kernel ISA: {sorted(ops - allowed)}"
    # second audit on the same listing: no kernel keeps anything in scratch memory (a register array indexed by a run-time
    # value ends up there — the chain walk of the first register version of the agent phases did: 2.3 us per step)
    # The builds that ask for the register budget of 8 wavefronts per SIMD (amdgpu_waves_per_eu) may spill a few dwords INSIDE
    # the rare delivery / reset path (the 128-bit PCG64 multiply needs more registers than the rest of the kernel has left):
    # every scratch instruction of such a kernel has to sit next to the PCG multiplier constant, and the frame stays tiny.
    lines = out.read_text().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN2rw17rware_step_kernel.*:\s", l)]
    assert starts
    n_scratch_kernels = 0
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        body = lines[a:b]
        size = [int(m.group(1)) for l in body for m in [re.match(r"^; ScratchSize: (\d+)", l)] if m]
        if not size or size[0] == 0:
            continue
        n_scratch_kernels += 1
        assert size[0] <= 64, f"{lines[a][:120]}: {size[0]} bytes of scratch per lane"
        pcg = [i for i, l in enumerate(body) if "0x4385df64" in l]   # low word of the PCG64 multiplier: only the RNG draw has it
        spills = [i for i, l in enumerate(body) if re.match(r"^\s*scratch_(load|store)", l)]
        assert pcg and spills and all(min(abs(i - j) for j in pcg) < 400 for i in spills), \
            f"{lines[a][:120]}: scratch traffic outside the rare RNG path"
    assert n_scratch_kernels <= 8, n_scratch_kernels
    # third audit: no register array indexed at run time (s_set_gpr_idx_on / v_movrel*) — that is what a table-driven loop turns
    # into when it stops being fully unrolled (round 4: one run-time `continue` in the stage-in's DMA job loop made the
    # headline step 6.1 -> 9.0 us; the compiler said nothing)
    dyn = re.findall(r"^\s*(s_set_gpr_idx_on|v_movrel[sd]*_b32)\b", out.read_text(), flags=re.M)
    assert not dyn, f"{len(dyn)} run-time-indexed register accesses in the exact-shape kernels"


def test_runtime_specialisation_compiles_without_a_device(tmp_path, monkeypatch):
    """rw_jit_probe: the compile half of what rw_create does for a shape with no ahead-of-time exact-shape kernel — the device
    headers embedded in the library, hipRTC, the disk cache — needs no GPU.  The shapes of three golden fixtures that run the
    generic kernel otherwise (a `layout=` string, column_height 5 with sensor_range 5, sensor_range 3) compile; the second
    request is a cache hit; an impossible shape fails with hipRTC's message, not a crash."""
    if not any(os.path.exists(p) for p in ("/opt/rocm/lib/libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7")):
        pytest.skip("no hipRTC on this box")
    monkeypatch.setenv("RWARE_JIT_CACHE", str(tmp_path))
    shapes = [dict(sensor_range=1, H=7, W=7, N=3, Q=3, S=20, E=16), dict(sensor_range=5, H=14, W=10, N=12, Q=12, S=60, E=8, nt=0),
              dict(sensor_range=3, H=20, W=10, N=3, Q=3, S=80, E=16)]
    for sh in shapes:
        n, log = _capi.jit_probe(**sh)
        assert n > 10000 and "compiled in" in log, log
        n2, log2 = _capi.jit_probe(**sh)
        assert n2 == n and log2.startswith("loaded "), log2
    assert len(list(tmp_path.glob("*.hsaco"))) == len(shapes)
    n, log = _capi.jit_probe(sensor_range=1, H=11, W=10, N=9, Q=9, S=32, E=4)       # a 4-env shelf chunk of 440 bytes: no whole DMA pieces
    assert n == -1 and "16-byte granular" in log, log


def test_runtime_specialisation_without_hiprtc_fails_softly(tmp_path):
    """A box without libhiprtc is the documented fall-back case (the generic kernel runs): the loader must come back with a message,
    not crash — ADVICE r4: dlerror() was called twice in one expression, the second call returns NULL, std::string + NULL.  The
    library handle is process-wide, so the no-library road is taken in a fresh interpreter (RWARE_JIT_LIBRARY: load exactly this)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from rware_amd import _capi; "
            "n, log = _capi.jit_probe(sensor_range=1, H=7, W=7, N=3, Q=3, S=20, E=16); print(n); print(log)" % ROOT)
    env = dict(os.environ, RWARE_JIT_LIBRARY=str(tmp_path / "no_such_libhiprtc.so"), RWARE_JIT_CACHE=str(tmp_path), RWARE_JIT_NO_CACHE="1")
    pr = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, (pr.returncode, pr.stderr[-800:])
    out = pr.stdout.strip().splitlines()
    assert out[0] == "-1" and "hipRTC not available" in out[1], out


def test_runtime_specialisation_keeps_no_cache_in_shared_directories(tmp_path):
    """Without HOME (and without RWARE_JIT_CACHE) there is no directory that is this user's alone: nothing is cached; a cache directory
    that others may write to is neither read nor written (a pre-planted code object would run inside the engine's process)."""
    import subprocess
    import sys
    if not any(os.path.exists(p) for p in ("/opt/rocm/lib/libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7")):
        pytest.skip("no hipRTC on this box")
    code = ("import sys; sys.path.insert(0, %r); from rware_amd import _capi; "
            "n, log = _capi.jit_probe(sensor_range=1, H=7, W=7, N=3, Q=3, S=20, E=16); print(n); print(log)" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("HOME", "RWARE_JIT_CACHE")}
    pr = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-800:]
    out = pr.stdout.strip().splitlines()
    assert int(out[0]) > 10000 and "not cached" in out[1], out
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    pr = subprocess.run([sys.executable, "-c", code], env=dict(env, RWARE_JIT_CACHE=str(shared)), capture_output=True, text=True, timeout=600)
    out = pr.stdout.strip().splitlines()
    assert pr.returncode == 0 and int(out[0]) > 10000 and "not a private directory" in out[1], (out, pr.stderr[-400:])
    assert not list(shared.glob("*.hsaco"))


def test_header_and_c_example_compile_as_plain_c11():
    """include/rware_hip.h is a C header (the boundary a cgo / JNI / ctypes binding reads): it and examples/rware_c_example.c — the
    boundary used from plain C, no Python, no torch — go through `gcc -std=c11 -pedantic` without a diagnostic."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for args in (["-x", "c", os.path.join(root, "include", "rware_hip.h")], [os.path.join(root, "examples", "rware_c_example.c")]):
        out = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include")] + args,
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
