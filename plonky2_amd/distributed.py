"""Coset-sharded PolynomialBatch commit across the GPUs of one node -- a thin caller of the library's multi-GPU
entry points (include/p2hot.h "multi-GPU", plonky2_amd/csrc/host_multi.hpp).

The sharding, the pipelined coefficient exchange and the cap / digest exchange all live in libp2hot
(`p2hot_commit_sharded_dev`, `p2hot_group_commit`); this module only
  * restates the shard arithmetic (`ShardPlan`, testable without devices),
  * builds the communicator: RCCL inside the library when the job runs one process per GPU under
    torch.distributed's "nccl" backend (the unique id travels through torch.distributed), or the library's
    caller-supplied-transport hook bound to torch.distributed/gloo in the CPU tests,
  * owns the device buffers of a rank (`ShardedCommit`) and wraps the single-process mode (`GroupCommit`).

The reference has no multi-device mode.  Design: the rate-1/B LDE is B independent coset transforms and coset j is the
contiguous row block bitrev(j) of the committed order; with B = 8 and cap_height = 4 a block is two whole cap subtrees,
so rank r owns rows [r*N/G, (r+1)*N/G) and runs the LDE, the leaf sponge and the Merkle levels of its rows with no
data-path exchange.  Exchanges: the coefficients after the column-sharded iNTT (W*n*8 bytes, pipelined in column
chunks), the cap entries, and -- optionally -- the digest slices (`gather_digests`; a Merkle path below the cap never
leaves the cap subtree of its leaf, so the owner of a row can always serve its path: `owner`, `prove_local`).
"""
import ctypes as C
import os

import numpy as np

from . import _lib

COSET_SHIFT = 14293326489335486720


class ShardPlan:
    """Pure arithmetic of the sharding (mirrors shard_plan() of csrc/host_multi.hpp; testable without devices)."""

    def __init__(self, W, log_n, rate_bits, cap_height, world):
        if world < 1 or world & (world - 1):
            raise ValueError("world size must be a power of two")
        self.W, self.log_n, self.rate_bits, self.cap_height, self.world = W, log_n, rate_bits, cap_height, world
        self.n = 1 << log_n
        self.N = self.n << rate_bits
        self.log_N = log_n + rate_bits
        if cap_height > self.log_N:
            raise ValueError("cap_height > log2(N) (merkle_tree.rs:195-200)")
        self.rows_per_rank = self.N // world
        sub_leaves = self.N >> cap_height
        if world > self.N:
            raise ValueError("world size %d exceeds the %d LDE rows" % (world, self.N))
        # more ranks than LDE cosets: every coset is split into 2^sub_bits sub-cosets of H_n, one per rank (csrc/host_multi.hpp)
        self.sub_bits = max(0, log_n - (self.rows_per_rank.bit_length() - 1))
        if world > (1 << cap_height) or self.rows_per_rank % sub_leaves:
            raise ValueError("world size %d exceeds the %d cap subtrees" % (world, 1 << cap_height))
        self.cols_per_rank = -(-W // world) if W else 0
        self.num_digests = 2 * (self.N - (1 << cap_height))
        self.digests_per_rank = self.num_digests // world
        self.cap_per_rank = (1 << cap_height) // world

    def rows(self, rank):
        return rank * self.rows_per_rank, self.rows_per_rank

    def columns(self, rank):
        c0 = min(self.W, rank * self.cols_per_rank)
        return c0, min(self.W, c0 + self.cols_per_rank)

    def cosets(self, rank):
        """natural coset indices j (points g * w_N^(B'*q + j), B' = 2^(rate_bits + sub_bits) cosets of size n >> sub_bits) whose
        rows this rank owns"""
        np_ = self.n >> self.sub_bits
        b0 = rank * self.rows_per_rank // np_
        nb = self.rows_per_rank // np_
        rb = self.rate_bits + self.sub_bits
        return [int(format(b, "0%db" % rb)[::-1], 2) if rb else 0 for b in range(b0, b0 + nb)]


def _gloo_transport(dist, rank):
    """p2hot_allgather_fn over torch.distributed with HOST buffers (the CPU test tier: the kernel emulator's "device"
    memory is host memory).  Rank r's slice is `nbytes` bytes at base + offsets[r]."""
    import torch

    def fn(_user, d_base, offsets, world, nbytes, _stream):
        try:
            view = lambda r: np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(d_base + offsets[r]))  # noqa: E731
            outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(outs, torch.from_numpy(view(rank).copy()))
            for r in range(world):
                if r != rank:
                    view(r)[:] = outs[r].numpy()
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            print("p2hot transport callback failed:", repr(e), flush=True)
            return 1
    return _lib.ALLGATHER_FN(fn)


class _DevBytes:
    """a raw device range as a __cuda_array_interface__ object (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _torch_transport(dist, rank):
    """p2hot_allgather_fn over torch.distributed with DEVICE buffers (backend "nccl" = torch's own RCCL communicator): the
    fallback when the library cannot bind RCCL itself.  Synchronous: waits for the library's communication stream, runs
    the all-gather on torch's, waits for it -- correct, but without the overlap of the in-library path."""
    import torch

    def fn(_user, d_base, offsets, world, nbytes, stream):
        try:
            if stream:
                torch.cuda.ExternalStream(stream).synchronize()
            else:
                torch.cuda.synchronize()
            slots = [torch.as_tensor(_DevBytes(d_base + offsets[r], nbytes), device="cuda") for r in range(world)]
            dist.all_gather(slots, slots[rank].clone())
            torch.cuda.synchronize()
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            print("p2hot transport callback failed:", repr(e), flush=True)
            return 1
    return _lib.ALLGATHER_FN(fn)


def _gloo_device_transport(dist, rank):
    """p2hot_allgather_fn over a gloo process group with DEVICE buffers staged through the host: lets several ranks share
    one GPU (RCCL refuses a repeated device), which is how the multi-process bench flow is exercised on a one-GPU box."""
    import torch

    def fn(_user, d_base, offsets, world, nbytes, stream):
        try:
            torch.cuda.synchronize()
            slots = [torch.as_tensor(_DevBytes(d_base + offsets[r], nbytes), device="cuda") for r in range(world)]
            outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(outs, slots[rank].cpu())
            for r in range(world):
                if r != rank:
                    slots[r].copy_(outs[r])
            torch.cuda.synchronize()
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            print("p2hot transport callback failed:", repr(e), flush=True)
            return 1
    return _lib.ALLGATHER_FN(fn)


class Communicator:
    """p2hot_comm: this rank's end of the library's exchange.  transport: "rccl" (inside libp2hot, one process per GPU),
    "torch" / "gloo" (caller-supplied hook over torch.distributed with device / host buffers) or None = pick by the process
    group's backend (P2HOT_TRANSPORT overrides): "rccl" under "nccl" when every rank can bind librccl, else "torch"."""

    def __init__(self, engine, rank, world, dist=None, transport=None):
        self.engine, self.rank, self.world = engine, rank, world
        self._cb = None
        h = C.c_void_p()
        if transport is None:
            transport = os.environ.get("P2HOT_TRANSPORT") or None
        if transport is None:
            transport = "none" if world == 1 or dist is None else ("rccl" if dist.get_backend() == "nccl" else "gloo")
            if transport == "rccl":  # can every rank bind RCCL inside the library?  (agreed on, so nobody waits alone)
                import torch
                probe = np.zeros(128, dtype=np.uint8)
                ok = torch.tensor([1 if engine.lib.p2hot_comm_unique_id(probe.ctypes.data) == 0 else 0], device=engine.mem.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    transport = "torch"
        self.transport = transport
        if transport == "rccl":
            import torch
            uid = np.zeros(128, dtype=np.uint8)
            if rank == 0:
                engine.check(engine.lib.p2hot_comm_unique_id(uid.ctypes.data))
            if world > 1:  # the launcher's job: hand rank 0's id to everybody
                dev = engine.mem.device if dist.get_backend() == "nccl" else "cpu"
                t = torch.from_numpy(uid).to(dev)
                dist.broadcast(t, src=0)
                uid = t.cpu().numpy()
            engine.check(engine.lib.p2hot_comm_create_rccl(engine.ctx, rank, world, uid.ctypes.data, C.byref(h)))
        else:
            on_gpu = getattr(getattr(engine.mem, "device", None), "type", None) == "cuda"
            self._cb = ((_gloo_device_transport if on_gpu else _gloo_transport)(dist, rank) if transport == "gloo"
                        else _torch_transport(dist, rank) if transport == "torch"
                        else _lib.ALLGATHER_FN(lambda *a: 0) if transport == "null"   # tools/rank_work.py: nothing is exchanged
                        else _lib.ALLGATHER_FN(lambda *a: 1))
            engine.check(engine.lib.p2hot_comm_create_callback(engine.ctx, rank, world, self._cb, None, C.byref(h)))
        self._h = h

    def selftest(self, nbytes=1 << 20):
        """collective preflight: a pattern slice per rank all-gathered and checked on every rank (p2hot_comm_selftest)"""
        if self.world > 1 or self.transport == "rccl":
            self.engine.check(self.engine.lib.p2hot_comm_selftest(self._h, nbytes))

    def rccl_info(self):
        """which RCCL the library is bound to in this process (p2hot_rccl_info): {"path", "version"} or None"""
        return rccl_info(self.engine.lib)

    @property
    def exchange(self):
        """how equal slices travel on the RCCL transport: "allgather" (ncclAllGather) or "broadcast" (grouped ncclBroadcast);
        decided by the selftest's micro-timing unless P2HOT_EXCHANGE pins it (include/p2hot.h, p2hot_comm_exchange_mode)"""
        if self.transport != "rccl" or not self._h:
            return None
        return {0: "broadcast", 1: "allgather"}.get(self.engine.lib.p2hot_comm_exchange_mode(self._h))

    def close(self):
        if self._h and getattr(self.engine, "_ctx", None):
            self.engine.lib.p2hot_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedCommit:
    """from_values / from_coeffs over `world` ranks, one process per GPU.  Buffers are allocated once and reused."""

    def __init__(self, engine, W, log_n, rate_bits, cap_height, is_values=True, rank=0, world=1, dist=None,
                 want_leaves=False, pipeline_chunks=None, gather_digests=True, transport=None, comm=None):
        """comm: an existing Communicator of the same (engine, rank, world) to reuse (one RCCL communicator serves every shape)"""
        self.eng, self.dist, self.rank, self.world = engine, dist, rank, world
        if pipeline_chunks is None:
            pipeline_chunks = 1 if os.environ.get("P2HOT_SYNC_COLLECTIVES") == "1" else 8
        self.pipeline_chunks = pipeline_chunks
        self.gather_digests = gather_digests
        self.plan = p = ShardPlan(W, log_n, rate_bits, cap_height, world)
        self.is_values = is_values
        mem = engine.mem
        self.column_range = p.columns(rank)
        self.row_begin, self.row_count = p.rows(rank)
        self.comm = comm if comm is not None else Communicator(engine, rank, world, dist, transport)
        # coefficient buffer, padded to world * cols_per_rank columns so every rank's slot has the same size
        self.coeffs_all = mem.empty(max(1, world * p.cols_per_rank), p.n)
        self.lde = mem.empty(max(W, 1), self.row_count)
        self.leaves = mem.empty(self.row_count, W) if want_leaves else None
        self.digests = mem.zeros(max(1, p.num_digests), 4)
        self.cap = mem.zeros(1 << cap_height, 4)

    def run(self, cols_local):
        """cols_local: device [c1 - c0][n] -- this rank's columns (values on H_n, or coefficients).
        Returns dict(coeffs [W][n], lde [W][rows of this rank], digests, cap), all device buffers;
        digests / cap are the FULL tree's arrays on every rank (with gather_digests=False only this rank's slice
        [rank * digests_per_rank, (rank + 1) * digests_per_rank) of `digests` is filled; cap is always complete)."""
        eng, p = self.eng, self.plan
        W = p.W
        stride = cols_local.shape[1] if (cols_local is not None and cols_local.ndim == 2 and cols_local.shape[0]) else p.n
        eng.check(eng.lib.p2hot_commit_sharded_dev(
            eng.ctx, self.comm._h, eng.ptr(cols_local) if cols_local is not None else None, stride, W, p.log_n, p.rate_bits,
            p.cap_height, 1 if self.is_values else 0, 1 if self.gather_digests else 0, self.pipeline_chunks,
            eng.ptr(self.coeffs_all), eng.ptr(self.lde), self.row_count, eng.ptr(self.leaves), eng.ptr(self.digests),
            eng.ptr(self.cap)))
        return {"coeffs": self.coeffs_all[:W], "lde": self.lde[:W], "leaves": self.leaves,
                "digests": self.digests[:p.num_digests], "cap": self.cap}

    def owner(self, leaf_index):
        """rank whose rows (and therefore whose digest slice) contain leaf `leaf_index` of the committed order"""
        return int(leaf_index) // self.plan.rows_per_rank

    def prove_local(self, leaf_indices):
        """(rows [m][W], Merkle paths [m][log2(N) - cap_height][4]) for leaves THIS rank owns, from its LDE block
        and its digest slice (MerkleTree::get + merkle_tree_prove, merkle_tree.rs:227, :151-190)."""
        eng, p = self.eng, self.plan
        idx = np.asarray(leaf_indices, dtype=np.uint64).reshape(-1)
        if any(self.owner(i) != self.rank for i in idx):
            raise ValueError("prove_local: a requested leaf belongs to another rank")
        layers = p.log_N - p.cap_height
        paths = eng.mem.zeros(max(1, len(idx)), max(layers, 1), 4)
        d_idx = eng.dev(idx if len(idx) else np.zeros(1, dtype=np.uint64))
        eng.check(eng.lib.p2hot_merkle_paths_dev(eng.ctx, eng.ptr(self.digests), p.log_N, p.cap_height, eng.ptr(d_idx),
                                                 len(idx), eng.ptr(paths)))
        rows = eng.gather_rows(self.lde[:p.W], idx - np.uint64(self.row_begin)) if len(idx) and p.W else \
            np.zeros((len(idx), p.W), dtype=np.uint64)
        return eng.host(rows), eng.host(paths)[:len(idx), :layers]


class GroupCommit:
    """p2hot_group: ONE process driving `n_gpus` devices -- what a patched plonky2 (a single Rust process) uses.
    `devices` may repeat a device id (one-GPU boxes, the kernel emulator): the ranks then exchange by copies."""

    def __init__(self, lib, n_gpus, devices=None):
        self.lib = lib
        h = C.c_void_p()
        dv = (C.c_int * n_gpus)(*(devices if devices is not None else range(n_gpus)))
        rc = lib.p2hot_group_create(n_gpus, dv, C.byref(h))
        if rc != _lib.OK:
            raise _lib.P2HotError(rc, "p2hot_group_create failed")
        self._h, self.n_gpus = h, n_gpus

    @property
    def uses_rccl(self):
        return bool(self.lib.p2hot_group_uses_rccl(self._h))

    @property
    def exchange(self):
        """how equal slices travel over RCCL: "allgather" (ncclAllGather) or "broadcast" (grouped ncclBroadcast), see include/p2hot.h"""
        return {0: "broadcast", 1: "allgather"}.get(self.lib.p2hot_group_exchange_mode(self._h), "none")

    def _check(self, rc):
        if rc != _lib.OK:
            raise _lib.P2HotError(rc, self.lib.p2hot_group_last_error(self._h).decode())

    def commit(self, cols, rate_bits, cap_height, is_values=True, want_leaves=False, want_digests=True, pipeline_chunks=8,
               by_columns=False):
        """cols: host [W][n].  Returns dict(coeffs, leaves, digests, cap) host arrays + an opener for rows / paths.
        by_columns: the column-sharded mode (P2HOT_SHARD_COLUMNS: whole-column LDEs + an all-to-all of the LDE matrix), a second
        partition with 2^rate_bits times the traffic; the default coset mode covers world > 2^rate_bits by sub-cosets."""
        cols = np.ascontiguousarray(np.asarray(cols, dtype=np.uint64))
        W, n = cols.shape
        log_n = int(n).bit_length() - 1
        N, ncap = n << rate_bits, 1 << cap_height
        ptrs = (C.c_void_p * max(W, 1))(*[cols[c].ctypes.data for c in range(W)])
        coeffs = np.zeros((W, n), dtype=np.uint64)
        leaves = np.zeros((N, W), dtype=np.uint64) if want_leaves else None
        digests = np.zeros((max(0, 2 * (N - ncap)), 4), dtype=np.uint64) if want_digests else None
        cap = np.zeros((ncap, 4), dtype=np.uint64)
        bh = C.c_void_p()
        self._check(self.lib.p2hot_group_commit(
            self._h, ptrs, W, log_n, rate_bits, cap_height, 1 if is_values else 0, 1 if by_columns else 0, pipeline_chunks,
            coeffs.ctypes.data,
            leaves.ctypes.data if want_leaves else None, digests.ctypes.data if want_digests else None, cap.ctypes.data,
            C.byref(bh)))
        lib, group = self.lib, self

        def open_(leaf_idx):
            idx = np.ascontiguousarray(np.asarray(leaf_idx, dtype=np.uint64).reshape(-1))
            layers = log_n + rate_bits - cap_height
            rows = np.zeros((len(idx), W), dtype=np.uint64)
            paths = np.zeros((len(idx), layers, 4), dtype=np.uint64)
            group._check(lib.p2hot_sharded_batch_open(bh, idx.ctypes.data, len(idx), rows.ctypes.data, paths.ctypes.data))
            return rows, paths

        def free():
            lib.p2hot_sharded_batch_free(bh)
        return {"coeffs": coeffs, "leaves": leaves, "digests": digests, "cap": cap, "open": open_, "free": free, "handle": bh,
                "W": W, "degree_log": log_n}

    def engine0(self):
        """an Engine-like view of rank 0's context (for the Challenger of a group proof)"""
        from .engine import Engine
        eng = Engine.__new__(Engine)
        eng.lib, eng._ctx, eng._stream = self.lib, C.c_void_p(self.lib.p2hot_group_ctx(self._h, 0)), None

        class _NoStream:
            def stream(self_inner):
                return None
        eng.mem = _NoStream()
        eng.close = lambda: None          # the group owns the context
        return eng

    def eval_openings(self, commits, points):
        """OpeningSet::new over sharded oracles (p2hot_group_eval_openings): list of [n_points][W][2] per oracle"""
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.uint64).reshape(-1, 2))
        hs = (C.c_void_p * len(commits))(*[c["handle"] for c in commits])
        total = sum(c["W"] for c in commits)
        flat = np.zeros(max(1, 2 * len(pts) * total), dtype=np.uint64)
        self._check(self.lib.p2hot_group_eval_openings(self._h, hs, len(commits), pts.ctypes.data, len(pts), flat.ctypes.data))
        out, off = [], 0
        for c in commits:
            cnt = 2 * len(pts) * c["W"]
            out.append(flat[off:off + cnt].reshape(len(pts), c["W"], 2))
            off += cnt
        return out

    def prove_openings(self, batches, commits, challenger, rate_bits, cap_height, reduction_arity_bits, proof_of_work_bits,
                       num_query_rounds):
        """p2hot_group_prove_openings: flat proof buffers as a dict (layout: include/p2hot.h, p2hot_fri_proof)"""
        arity = [int(a) for a in reduction_arity_bits]
        ab = (C.c_uint * max(len(arity), 1))(*arity)
        fp = _lib.FriParams(rate_bits, cap_height, proof_of_work_bits, num_query_rounds, ab, len(arity), 0, 0, 0)
        keep, infos = [], (_lib.FriBatchInfo * max(len(batches), 1))()
        for k, (point, polys) in enumerate(batches):
            oi = (C.c_uint32 * max(len(polys), 1))(*[o for o, _ in polys])
            pi = (C.c_uint32 * max(len(polys), 1))(*[p for _, p in polys])
            keep += [oi, pi]
            infos[k].point[0], infos[k].point[1] = int(point[0]), int(point[1])
            infos[k].oracle_index, infos[k].poly_index, infos[k].n_polys = oi, pi, len(polys)
        hs = (C.c_void_p * len(commits))(*[c["handle"] for c in commits])
        lay = _lib.FriProofLayout()
        rc = self.lib.p2hot_group_fri_proof_sizes(hs, len(commits), C.byref(fp), C.byref(lay))
        if rc != _lib.OK:
            raise _lib.P2HotError(rc, "inconsistent FRI parameters")
        names = ("caps", "final_poly", "initial_leaves", "initial_paths", "step_evals", "step_paths")
        bufs = {k: np.zeros(max(1, getattr(lay, k + "_words")), dtype=np.uint64) for k in names}
        qidx = np.zeros(max(1, num_query_rounds), dtype=np.uint64)
        proof = _lib.FriProof(bufs["caps"].ctypes.data, bufs["final_poly"].ctypes.data, 0, qidx.ctypes.data,
                              bufs["initial_leaves"].ctypes.data, bufs["initial_paths"].ctypes.data,
                              bufs["step_evals"].ctypes.data, bufs["step_paths"].ctypes.data)
        self._check(self.lib.p2hot_group_prove_openings(self._h, infos, len(batches), hs, len(commits), challenger._h,
                                                        C.byref(fp), C.byref(proof)))
        out = {k: bufs[k][:getattr(lay, k + "_words")] for k in names}
        out["pow_witness"], out["query_indices"] = int(proof.pow_witness), [int(x) for x in qidx[:num_query_rounds]]
        return out

    def close(self):
        if self._h:
            self.lib.p2hot_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_info(lib):
    """p2hot_rccl_info: the RCCL file libp2hot bound in THIS process and its ncclGetVersion code -- PyTorch's bundled copy when
    torch loaded it first (RTLD_NOLOAD), /opt/rocm's by path otherwise; None when no RCCL binds"""
    buf = C.create_string_buffer(1024)
    ver = C.c_int(0)
    if lib.p2hot_rccl_info(buf, len(buf), C.byref(ver)) != 0:
        return None
    return {"path": buf.value.decode(errors="replace"), "version": int(ver.value)}
