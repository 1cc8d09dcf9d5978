import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P = 0xFFFFFFFF00000001


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (test infrastructure)."""
    from oracle import p2oracle
    p2oracle.set_num_threads(p2oracle.usable_cores())  # do not oversubscribe a cgroup-limited box
    return p2oracle


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def emu():
    """Engine over the test-only kernel-source emulator (tests/emu): same kernels, g++, no GPU."""
    from tests.emu_backend import emu_engine
    return emu_engine()


@pytest.fixture(scope="session")
def gpu():
    """The product engine: libp2hot.so (HIP, gfx950) on cuda:0 through the C ABI."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X (run with -m gpu on the GPU box)")
    import __graft_entry__ as ge
    ge.build_product()
    from plonky2_amd import Engine
    return Engine(0)


BACKENDS = ["emu", pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def eng(request):
    """Every parity test runs on the emulator here (CPU tier) and on the MI355X (-m gpu)."""
    return request.getfixturevalue(request.param)


def rand_field(rng, *shape, noncanonical=False):
    """uniform field elements; with noncanonical=True some entries are representatives in [P, 2^64)"""
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    if noncanonical:
        m = rng.random(shape) < 0.05
        small = rng.integers(0, 2**32 - 1, size=shape, dtype=np.uint64)
        a = np.where(m, small + np.uint64(P), a)
    return a


def pytest_sessionfinish(session, exitstatus):
    """P2HOT_EMU_ASM=1 (the whole CPU tier through the emulator's instruction interpreter): any hazard / clobber / definition
    report of the interpreter fails the run, and the run must have interpreted something"""
    if os.environ.get("P2HOT_EMU_ASM") != "1":
        return
    import ctypes as C
    from tests import emu_backend
    lib = emu_backend._lib_cache
    if lib is None:
        return
    lib.p2hot_emu_asm_stats.restype = C.c_ulonglong
    lib.p2hot_emu_asm_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_char_p, C.c_size_t]
    blocks, errors, buf = C.c_ulonglong(), C.c_ulonglong(), C.create_string_buffer(1024)
    n = lib.p2hot_emu_asm_stats(C.byref(blocks), C.byref(errors), buf, 1024)
    print("\n[gcn_asm] %d instructions in %d asm blocks interpreted, %d reports %s" % (n, blocks.value, errors.value, buf.value.decode()))
    if errors.value:
        session.exitstatus = 1
