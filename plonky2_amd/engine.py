"""Engine: one libp2hot context bound to one GPU, plus the device-memory plumbing.

PyTorch is used only as plumbing (device allocations, H2D/D2H copies, the current HIP stream and,
in plonky2_amd.distributed, torch.distributed over RCCL).  All arithmetic happens in the HIP
kernels of libp2hot; there is no CPU fallback -- constructing an Engine without a visible GPU or
without the built library raises.
"""
import ctypes as C

import numpy as np

from . import _lib

P = 0xFFFFFFFF00000001
COSET_SHIFT = 14293326489335486720


class TorchMemory:
    """Device buffers as int64 CUDA tensors (bit patterns are the u64 field elements)."""

    def __init__(self, device_index=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("plonky2_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.torch = torch
        self.device_index = device_index
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)

    def empty(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.int64, device=self.device)

    def zeros(self, *shape):
        return self.torch.zeros(shape, dtype=self.torch.int64, device=self.device)

    def from_host(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return self.torch.from_numpy(a.view(np.int64)).to(self.device)

    def to_host(self, t):
        return t.detach().cpu().contiguous().numpy().view(np.uint64)

    def is_buffer(self, x):
        return isinstance(x, self.torch.Tensor)

    def ptr(self, t):
        return t.data_ptr()

    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def as_torch(self, t):
        """the buffer as a torch tensor for torch.distributed collectives"""
        return t

    def collective_fence(self):
        """libp2hot enqueues on the current torch stream, and so do the collectives: nothing to do"""


class Engine:
    def __init__(self, device_index=0, lib=None, memory=None):
        self.lib = lib if lib is not None else _lib.product()
        self.mem = memory if memory is not None else TorchMemory(device_index)
        ctx = C.c_void_p()
        self._stream = self.mem.stream()
        rc = self.lib.p2hot_ctx_create(device_index, self._stream, C.byref(ctx))
        self._ctx = ctx
        if rc != _lib.OK:
            msg = self.lib.p2hot_last_error(ctx).decode() if ctx else "context creation failed"
            if ctx:
                self.lib.p2hot_ctx_destroy(ctx)
            self._ctx = None
            raise _lib.P2HotError(rc, msg)

    @property
    def ctx(self):
        """The library context, re-bound to the memory backend's CURRENT stream if that changed since the last call
        (torch.cuda.stream(...) blocks): torch copies / collectives and libp2hot kernels then stay ordered on one
        stream.  Every library call takes `eng.ctx` as its first argument, so this is the one place to do it."""
        if self._ctx:
            st = self.mem.stream()
            if st != self._stream:
                rc = self.lib.p2hot_ctx_set_stream(self._ctx, st)  # synchronises the old stream first
                if rc != _lib.OK:
                    raise _lib.P2HotError(rc, self.lib.p2hot_last_error(self._ctx).decode())
                self._stream = st
        return self._ctx

    @ctx.setter
    def ctx(self, value):
        self._ctx = value

    # -- plumbing
    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.p2hot_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != _lib.OK:
            raise _lib.P2HotError(rc, self.lib.p2hot_last_error(self._ctx).decode())

    def sync(self):
        self.check(self.lib.p2hot_ctx_sync(self.ctx))

    def dev(self, x):
        """host ndarray (any shape, uint64) -> device buffer; device buffers pass through"""
        return x if self.mem.is_buffer(x) else self.mem.from_host(x)

    def host(self, x):
        if self.mem.is_buffer(x):
            return self.mem.to_host(x)
        if hasattr(x, "degree_log") and callable(getattr(x, "host", None)):  # fri.oracle.DeviceColumns
            return x.host()
        return np.asarray(x, dtype=np.uint64)

    def ptr(self, x):
        return C.c_void_p(self.mem.ptr(x)) if x is not None else None

    def profile(self, on=True):
        self.check(self.lib.p2hot_profile_enable(self.ctx, 1 if on else 0))

    def profile_results(self, reset=True):
        """{kernel: {"ms": total, "launches": count}} measured with HIP events on the launch stream"""
        import json
        return json.loads(self.lib.p2hot_profile_json(self.ctx, 1 if reset else 0).decode())

    def num_digests(self, log_leaves, cap_height):
        return self.lib.p2hot_num_digests(log_leaves, cap_height)

    # -- primitives on device buffers shaped [batch][n]
    def fft(self, buf, log_n):
        batch, stride = buf.shape
        self.check(self.lib.p2hot_fft_dev(self.ctx, self.ptr(buf), batch, stride, log_n))
        return buf

    def ifft(self, buf, log_n):
        batch, stride = buf.shape
        self.check(self.lib.p2hot_ifft_dev(self.ctx, self.ptr(buf), batch, stride, log_n))
        return buf

    def coset_lde(self, coeffs, log_n, rate_bits, shift=COSET_SHIFT, row_begin=0, row_count=None):
        W, stride = coeffs.shape
        N = 1 << (log_n + rate_bits)
        if row_count is None:
            row_count = N - row_begin
        out = self.mem.empty(W, row_count)
        self.check(self.lib.p2hot_coset_lde_dev(self.ctx, self.ptr(coeffs), W, stride, log_n, rate_bits, shift,
                                                row_begin, row_count, self.ptr(out), row_count))
        return out

    def transpose(self, colmajor):
        W, rows = colmajor.shape
        out = self.mem.empty(rows, W)
        self.check(self.lib.p2hot_transpose_dev(self.ctx, self.ptr(colmajor), rows, W, rows, self.ptr(out)))
        return out

    def reverse_index_bits(self, buf, log_n):
        batch, stride = buf.shape
        out = self.mem.empty(batch, stride)
        self.check(self.lib.p2hot_reverse_index_bits_dev(self.ctx, self.ptr(buf), self.ptr(out), batch, stride, log_n))
        return out

    def poseidon_permute(self, states):
        count = states.shape[0]
        self.check(self.lib.p2hot_poseidon_permute_dev(self.ctx, self.ptr(states), count))
        return states

    def merkle(self, leaves, layout, W, log_leaves, cap_height, leaf_begin=0, leaf_count=None, digests=None, cap=None):
        """layout 0: leaves [W][leaf_count] column-major; layout 1: [leaf_count][W] row-major"""
        n_leaves = 1 << log_leaves
        if leaf_count is None:
            leaf_count = n_leaves - leaf_begin
        nd = self.num_digests(log_leaves, cap_height)
        if digests is None:
            digests = self.mem.zeros(max(nd, 1), 4)
        if cap is None:
            cap = self.mem.zeros(1 << cap_height, 4)
        stride = leaves.shape[1] if (layout == 0 and W) else 0
        self.check(self.lib.p2hot_merkle_dev(self.ctx, self.ptr(leaves) if W else None, layout, stride, W, log_leaves,
                                             cap_height, leaf_begin, leaf_count, self.ptr(digests), self.ptr(cap)))
        return digests[:nd], cap

    def gather_rows(self, colmajor, idx):
        W, stride = colmajor.shape
        if not self.mem.is_buffer(idx):  # host indices are validated here; device-resident ones by the kernel (p2hot_ctx_sync)
            idx = np.asarray(idx, dtype=np.uint64)
            if idx.size and int(idx.max()) >= stride:
                raise IndexError("gather_rows: row index %d out of range (%d rows)" % (int(idx.max()), stride))
        idx = self.dev(idx)
        m = idx.shape[0]
        out = self.mem.empty(m, W)
        self.check(self.lib.p2hot_gather_rows_dev(self.ctx, self.ptr(colmajor), stride, stride, W, self.ptr(idx), m,
                                                  self.ptr(out)))
        return out

    def commit(self, cols, log_n, rate_bits, cap_height, is_values, row_begin=0, row_count=None, want_leaves=False,
               digests=None, cap=None):
        """p2hot_commit_dev.  cols: device [W][n].  Returns dict of device buffers."""
        W, stride = cols.shape
        n = 1 << log_n
        log_N = log_n + rate_bits
        N = 1 << log_N
        if row_count is None:
            row_count = N - row_begin
        nd = self.num_digests(log_N, cap_height)
        coeffs = self.mem.empty(W, n) if is_values else cols
        lde = self.mem.empty(W, row_count)
        leaves = self.mem.empty(row_count, W) if want_leaves else None
        if digests is None:
            digests = self.mem.zeros(max(nd, 1), 4)
        if cap is None:
            cap = self.mem.zeros(1 << cap_height, 4)
        self.check(self.lib.p2hot_commit_dev(
            self.ctx, self.ptr(cols), stride, W, log_n, rate_bits, cap_height, 1 if is_values else 0, row_begin,
            row_count, self.ptr(coeffs), coeffs.shape[1] if W else n, self.ptr(lde), row_count, self.ptr(leaves),
            self.ptr(digests), self.ptr(cap)))
        return {"coeffs": coeffs, "lde": lde, "leaves": leaves, "digests": digests[:nd], "cap": cap}


_default = None


def default_engine():
    """Process-wide engine on the current CUDA device (created on first use)."""
    global _default
    if _default is None:
        import torch
        idx = torch.cuda.current_device() if torch.cuda.is_available() else 0
        _default = Engine(idx)
    return _default


def set_default_engine(engine):
    global _default
    _default = engine
