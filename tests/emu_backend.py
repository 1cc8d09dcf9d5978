"""TEST INFRASTRUCTURE ONLY: an Engine over the kernel-source emulator build (tests/emu).

The emulator compiles the very same plonky2_amd/csrc sources with g++ against a single-threaded
model of the HIP grid/block/__syncthreads semantics, so `pytest -m "not gpu"` can check kernel index
arithmetic, the host orchestration and the multi-rank sharding logic against the oracle in the
GPU-less build container.  "Device" buffers are numpy arrays.  Never imported by plonky2_amd.
"""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

from plonky2_amd import _lib
from plonky2_amd.engine import Engine

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "emu")
_SO = os.path.join(_DIR, "libp2hot_emu.so")


class DevArray(np.ndarray):
    """marks an ndarray as an emulated *device* buffer, so host arrays are always copied in (as on the GPU)"""


def _hip(lib, name, argtypes):
    """a C++ entry point of the emulated HIP runtime (tests/emu/hip_emu_rt.cpp), found by its mangled name"""
    import re
    global _syms
    if _syms is None:
        _syms = subprocess.run(["nm", "-D", "--defined-only", _SO], capture_output=True, text=True).stdout
    m = re.search(r"\b(_Z%d%s\w*)\b" % (len(name), name), _syms)
    f = getattr(lib, m.group(1))
    f.argtypes, f.restype = argtypes, C.c_int
    return f


_syms = None


class HostMemory:
    """"Device" buffers of the test tier.  They come from the emulated hipMalloc, so they carry the device that was current when
    they were made (the fake RCCL and the peer-copy rules see them), sit behind that device's page protection, and END at a guard
    page: a kernel of a *_dev entry point that indexes past the end of a caller's buffer faults with a named message instead of
    scribbling over numpy's heap.  Each block is returned to the emulated runtime when its last numpy view dies."""

    def __init__(self):
        lib = emu_lib()
        self._malloc = _hip(lib, "hipMalloc", [C.POINTER(C.c_void_p), C.c_size_t])
        self._free = _hip(lib, "hipFree", [C.c_void_p])
        self._sync = _hip(lib, "hipStreamSynchronize", [C.c_void_p])
        self._errstr = _hip(lib, "hipGetErrorString", [C.c_int])
        self._errstr.restype = C.c_char_p

    def _synchronize(self):
        """the emulated streams are queues (tests/emu/hip_emu_rt.cpp): work runs when somebody waits for it.  The engines of this
        tier sit on the null stream, so this is torch's `.cpu()` / current-stream synchronisation of the GPU tier"""
        rc = self._sync(None)
        if rc != 0:
            raise RuntimeError("emulated hipStreamSynchronize: %s" % self._errstr(rc).decode())

    def _block(self, shape, zero):
        words = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        if words == 0:
            return np.zeros(shape, dtype=np.uint64).view(DevArray)
        p = C.c_void_p()
        if self._malloc(C.byref(p), words * 8) != 0:
            raise MemoryError("emulated hipMalloc(%d)" % (words * 8))
        buf = (C.c_uint64 * words).from_address(p.value)
        weakref.finalize(buf, self._free, p.value)   # every numpy view keeps `buf` alive through its base chain
        a = np.frombuffer(buf, dtype=np.uint64).reshape(shape)
        if zero:
            a[...] = 0   # (mmap'ed pages are zero, but a recycled block need not be)
        return a.view(DevArray)

    def empty(self, *shape):
        return self._block(shape, False)

    def zeros(self, *shape):
        return self._block(shape, True)

    def from_host(self, a):
        a = np.asarray(a, dtype=np.uint64)
        d = self._block(a.shape, False)
        d[...] = a
        return d

    def to_host(self, t):
        self._synchronize()
        return np.array(t, dtype=np.uint64, copy=True).view(np.ndarray)

    def is_buffer(self, x):
        return isinstance(x, DevArray)

    def ptr(self, t):
        assert t.flags["C_CONTIGUOUS"]
        return t.ctypes.data

    def stream(self):
        return None

    def as_torch(self, t):
        import torch
        assert t.flags["C_CONTIGUOUS"]
        self._synchronize()
        return torch.from_numpy(t.view(np.ndarray).view(np.int64))  # shares memory

    def collective_fence(self):
        self._synchronize()


_lib_cache = None
# launches of up to this many permutations run the quad-cooperative / word-per-lane Poseidon kernels under the emulator
# (the product's defaults are 2^15 / 2^13; emulated cross-lane exchanges cost two context switches per lane)
EMU_TUNE_QUAD, EMU_TUNE_ROW = 512, 128


def emu_lib():
    global _lib_cache
    # every emulated context of the process (those p2hot_group_create makes too) starts from the emulator's thresholds:
    # all three Poseidon mappings run, the slow-to-emulate cooperative ones only on the smallest launches
    os.environ.setdefault("P2HOT_TUNE_QUAD", str(EMU_TUNE_QUAD))
    os.environ.setdefault("P2HOT_TUNE_ROW", str(EMU_TUNE_ROW))
    if _lib_cache is None:
        subprocess.check_call(["make", "-C", _DIR, "-s"])
        _lib_cache = _lib.load(_SO)
        assert _lib_cache.p2hot_is_emulated() == 1
        if os.environ.get("P2HOT_EMU_ASM") == "1":  # the whole tier through the instruction interpreter (slow: ~30x on Poseidon)
            _lib_cache.p2hot_emu_asm(1)
    return _lib_cache


def emu_engine():
    return Engine(0, lib=emu_lib(), memory=HostMemory())
