"""TEST INFRASTRUCTURE ONLY: an Engine over the kernel-source emulator build (tests/emu).

The emulator compiles the very same plonky2_amd/csrc sources with g++ against a single-threaded
model of the HIP grid/block/__syncthreads semantics, so `pytest -m "not gpu"` can check kernel index
arithmetic, the host orchestration and the multi-rank sharding logic against the oracle in the
GPU-less build container.  "Device" buffers are numpy arrays.  Never imported by plonky2_amd.
"""
import os
import subprocess

import numpy as np

from plonky2_amd import _lib
from plonky2_amd.engine import Engine

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "emu")
_SO = os.path.join(_DIR, "libp2hot_emu.so")


class DevArray(np.ndarray):
    """marks an ndarray as an emulated *device* buffer, so host arrays are always copied in (as on the GPU)"""


class HostMemory:
    def empty(self, *shape):
        return np.empty(shape, dtype=np.uint64).view(DevArray)

    def zeros(self, *shape):
        return np.zeros(shape, dtype=np.uint64).view(DevArray)

    def from_host(self, a):
        return np.array(a, dtype=np.uint64, order="C", copy=True).view(DevArray)

    def to_host(self, t):
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
        return torch.from_numpy(t.view(np.ndarray).view(np.int64))  # shares memory

    def collective_fence(self):
        pass


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
