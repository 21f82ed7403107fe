import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_SRC = os.path.join(ROOT, "tests", "csrc", "emu.cpp")
EMU_LIB = os.path.join(ROOT, "tests", "csrc", "libglvemu.so")
EMU64_SRC = os.path.join(ROOT, "tests", "csrc", "emu64.cpp")
EMU64_LIB = os.path.join(ROOT, "tests", "csrc", "libglvemu64.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def build_emulation():
    """g++ build of the CPU lane emulation of the wave engine (test infrastructure)."""
    srcs = [EMU_SRC] + [os.path.join(ROOT, "tts_amd", "csrc", h)
                        for h in ("wave_fft.h", "engine.h", "elementwise.h", "host_tables.h")]
    if os.path.exists(EMU_LIB) and all(os.path.getmtime(EMU_LIB) >= os.path.getmtime(s) for s in srcs):
        return EMU_LIB
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                           "-o", EMU_LIB, EMU_SRC])
    return EMU_LIB


def build_emulation64():
    """The same emulation with every float compiled as double (tests/csrc/emu64.cpp): shows that
    the engine's iteration is librosa's algorithm exactly, rounding aside."""
    srcs = [EMU64_SRC, EMU_SRC] + [os.path.join(ROOT, "tts_amd", "csrc", h)
                                   for h in ("wave_fft.h", "engine.h", "elementwise.h", "host_tables.h")]
    if os.path.exists(EMU64_LIB) and all(os.path.getmtime(EMU64_LIB) >= os.path.getmtime(s) for s in srcs):
        return EMU64_LIB
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                           "-o", EMU64_LIB, EMU64_SRC])
    return EMU64_LIB


@pytest.fixture(scope="session")
def emu():
    lib = ctypes.CDLL(build_emulation())
    return lib


@pytest.fixture(scope="session")
def emu64():
    return ctypes.CDLL(build_emulation64())


@pytest.fixture(scope="session")
def lj001():
    g = np.load(os.path.join(GOLDEN, "lj001_mel.npz"))
    return g["mel_db"], np.float32(g["mean"]), np.float32(g["std"])


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
