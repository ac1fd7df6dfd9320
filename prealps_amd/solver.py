"""Host-side mirror of the reference driver (examples/test_ecg_prealps_op.c:151-239)
on top of the C ABI: build the operator, factor the block-Jacobi preconditioner,
run the reverse-communication loop.  Everything numeric happens in
libprealps_hip.so on the GPU; numpy only carries host arrays in and out.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import lib as _l
from .lib import (ADAPT_BS, NO_BS_RED, ORTHODIR, ORTHODIR_FUSED, ORTHOMIN, CPLM_Mat_CSR_t,
                  CPLM_Mat_Dense_t, check, preAlps_ECG_t)


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@dataclass
class EcgResult:
    x: np.ndarray
    iters: int
    res: np.ndarray          # residual norm after every stopping test
    bs: np.ndarray           # block size after every stopping test
    final_res: float
    final_bs: int
    normb: float
    seconds: float = 0.0
    timers: dict = field(default_factory=dict)


class DistributedHooks:
    """Binds the library's two communication hooks to torch.distributed.

    backend "nccl" is RCCL on ROCm: the device buffers are wrapped zero-copy as
    torch tensors and reduced / exchanged in place.  With "gloo" (CPU tests and
    single-GPU rehearsals) the buffers are staged through host memory.
    """

    def __init__(self, L):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.L = torch, dist, L
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.staged = dist.get_backend() != "nccl"
        # everything the hooks do is queued on the library's own stream, so it is ordered
        # after the kernels that produced the buffers and before the ones that consume them
        self._streams = {}
        self._ar = _l.ALLREDUCE_FN(self._allreduce)
        self._ex = _l.EXCHANGE_FN(self._exchange)
        check(L.preAlps_hip_set_world(self.rank, self.size), "preAlps_hip_set_world")
        check(L.preAlps_hip_set_comm(self._ar, self._ex, None), "preAlps_hip_set_comm")

    def _stream(self):
        # the library switches to its side stream around the halo exchange
        h = int(self.L.preAlps_hip_get_stream())
        st = self._streams.get(h)
        if st is None:
            st = self._streams[h] = self.torch.cuda.ExternalStream(h)
        return st

    def _wrap(self, ptr, count):
        """Zero-copy float64 view of `count` doubles of device memory owned by the library."""
        torch = self.torch
        dev = torch.device("cuda", torch.cuda.current_device())
        st = torch._C._construct_storage_from_data_pointer(int(ptr), dev, 8 * int(count))
        t = torch.empty(0, dtype=torch.float64, device=dev).set_(st, 0, (int(count),), (1,))
        if t.data_ptr() != int(ptr):
            raise RuntimeError("could not alias library memory as a torch tensor")
        return t

    def _allreduce(self, ctx, ptr, count):
        try:
            with self.torch.cuda.stream(self._stream()):
                t = self._wrap(ptr, count)
                if self.staged:
                    h = t.cpu()
                    self.dist.all_reduce(h)
                    t.copy_(h)
                else:
                    self.dist.all_reduce(t)
            return 0
        except Exception as e:  # pragma: no cover - surfaced through the C error path
            print("all-reduce hook failed:", e)
            return 1

    def _exchange(self, ctx, send, send_counts, recv, recv_counts, peers, npeers):
        try:
            dist, torch = self.dist, self.torch
            ns = sum(send_counts[i] for i in range(npeers))
            nr = sum(recv_counts[i] for i in range(npeers))
            with torch.cuda.stream(self._stream()):
                ts_ = self._wrap(send, max(ns, 1))
                tr_ = self._wrap(recv, max(nr, 1))
                if self.staged:
                    hs, hr = ts_.cpu(), torch.empty(max(nr, 1), dtype=torch.float64)
                else:
                    hs, hr = ts_, tr_
                ops, so, ro = [], 0, 0
                for i in range(npeers):
                    p, sc, rc = peers[i], send_counts[i], recv_counts[i]
                    if sc:
                        ops.append(dist.P2POp(dist.isend, hs[so:so + sc], p))
                    if rc:
                        ops.append(dist.P2POp(dist.irecv, hr[ro:ro + rc], p))
                    so += sc
                    ro += rc
                for w in dist.batch_isend_irecv(ops) if ops else []:
                    w.wait()
                if self.staged and nr:
                    tr_[:nr].copy_(hr[:nr])
            return 0
        except Exception as e:  # pragma: no cover
            print("halo-exchange hook failed:", e)
            return 1


def bind_process_group(L, prefer=None):
    """Install the communication hooks of this process for the current torch.distributed
    group.  With the "nccl" backend the library's own RCCL binding is used (C, on the
    library stream: no host round trip); the torch.distributed hooks are the fallback
    and what "gloo" uses.  Returns (kind, keep_alive_object)."""
    import os
    import ctypes as C
    import torch.distributed as dist
    prefer = prefer or os.environ.get("PREALPS_COMM", "rccl")
    rank, size = dist.get_rank(), dist.get_world_size()
    if prefer == "rccl" and dist.get_backend() == "nccl":
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())

        def all_ok(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        # vote BEFORE the collective ncclCommInitRank: a rank that cannot load librccl or (rank 0)
        # cannot create the id must not leave the others blocked inside it
        buf = C.create_string_buffer(128)
        ok = L.preAlps_hip_rccl_available() == 0
        if ok and rank == 0:
            ok = L.preAlps_hip_rccl_unique_id(buf) == 0
        if not ok:
            print("[prealps_amd] native RCCL hooks unavailable on rank %d (%s)" % (rank, L.preAlps_hip_last_error()))
        if all_ok(ok):
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0)
            ok = L.preAlps_hip_rccl_init(box[0], rank, size) == 0
            ok = ok and L.preAlps_hip_comm_selftest() == 0
            if not ok:
                print("[prealps_amd] native RCCL binding failed on rank %d (%s)" % (rank, L.preAlps_hip_last_error()))
            # every rank must end up on the same binding: one failure sends all of them to the fallback
            if all_ok(ok):
                return "rccl", None
        if rank == 0:
            print("[prealps_amd] using the torch.distributed hooks on all ranks")
    hooks = DistributedHooks(L)
    check(L.preAlps_hip_comm_selftest(), "preAlps_hip_comm_selftest")
    return "torch." + dist.get_backend(), hooks


def partition_kway(rowptr, colind, nparts):
    """preAlps_hip_partition_kway: the library's stand-in for METIS_PartGraphKway (host code)."""
    L = _l.load()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    part = np.empty(len(rowptr) - 1, dtype=np.int32)
    check(L.preAlps_hip_partition_kway(len(part), _pi(rowptr), _pi(colind), int(nparts), _pi(part)),
          "preAlps_hip_partition_kway")
    return part


class EcgProblem:
    """One operator + one block-Jacobi preconditioner (both process-global in
    the library, as in the reference) and solves on them."""

    def __init__(self, rowptr, colind, val, nparts, part=None, scale=True, device=None,
                 distributed=False, use_torch_stream=False, partitioner=False, shard=None):
        """part: explicit partition vector; None = contiguous row blocks, or -- with
        partitioner=True -- the library's k-way graph partitioner (what preAlps_OperatorBuild
        uses where the reference calls METIS).  shard = (r, G): rehearse rank r of a G-process run in
        this one process (preAlps_hip_loopback)."""
        self.L = L = _l.load()
        import os
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
        check(L.preAlps_hip_init(dev), "preAlps_hip_init")
        self.hooks, self.comm_kind = None, "none"
        if distributed:
            import torch
            torch.cuda.set_device(dev)
            self.comm_kind, self.hooks = bind_process_group(L)
        if shard is not None:
            check(L.preAlps_hip_loopback(int(shard[0]), int(shard[1])), "preAlps_hip_loopback")
            self.comm_kind = "loopback %d/%d" % (shard[0], shard[1])
        if use_torch_stream:
            import torch
            check(L.preAlps_hip_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "preAlps_hip_set_stream")
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colind = np.ascontiguousarray(colind, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        self.N = len(rowptr) - 1
        p = None if part is None else np.ascontiguousarray(part, dtype=np.int32)
        if p is None and partitioner:
            p = partition_kway(rowptr, colind, nparts)
        check(L.preAlps_OperatorBuildFromCSR(self.N, _pi(rowptr), _pi(colind), _pd(val), int(nparts),
                                             None if p is None else _pi(p), 1 if scale else 0),
              "preAlps_OperatorBuildFromCSR")
        self._after_build()

    @classmethod
    def from_mtx(cls, path, nparts=None, device=0, partition=None):
        """preAlps_OperatorBuild(file, comm) like the reference driver.  partition: None = the
        library's graph partitioner (where the reference calls METIS), "contiguous" = row blocks."""
        import os
        self = cls.__new__(cls)
        self.L = L = _l.load()
        self.hooks, self.comm_kind = None, "none"
        check(L.preAlps_hip_init(device), "preAlps_hip_init")
        if nparts is not None:
            os.environ["PREALPS_NPARTS"] = str(nparts)
        if partition is None:
            os.environ.pop("PREALPS_PARTITION", None)
        else:
            os.environ["PREALPS_PARTITION"] = partition
        check(L.preAlps_OperatorBuild(path.encode(), 0x44000000), "preAlps_OperatorBuild")
        self._after_build()
        return self

    def _after_build(self):
        L = self.L
        M, m = C.c_int(), C.c_int()
        check(L.preAlps_OperatorGetSizes(C.byref(M), C.byref(m)), "preAlps_OperatorGetSizes")
        self.M, self.m = M.value, m.value
        self.N = self.M
        self.A = CPLM_Mat_CSR_t()
        check(L.preAlps_OperatorGetA(C.byref(self.A)), "preAlps_OperatorGetA")
        rp, n = C.POINTER(C.c_int)(), C.c_int()
        check(L.preAlps_OperatorGetRowPosPtr(C.byref(rp), C.byref(n)), "preAlps_OperatorGetRowPosPtr")
        self._rowpos_ptr, self._rowpos_n = rp, n.value
        self.rowpos = np.ctypeslib.as_array(rp, shape=(n.value,)).copy()
        cp, cn = C.POINTER(C.c_int)(), C.c_int()
        check(L.preAlps_OperatorGetColPosPtr(C.byref(cp), C.byref(cn)), "preAlps_OperatorGetColPosPtr")
        self._colpos_ptr, self._colpos_n = cp, cn.value
        pp, pn = C.POINTER(C.c_int)(), C.c_int()
        check(L.preAlps_OperatorGetPermPtr(C.byref(pp), C.byref(pn)), "preAlps_OperatorGetPermPtr")
        self.perm = np.ctypeslib.as_array(pp, shape=(pn.value,)).copy()
        self.nparts = L.preAlps_hip_nparts()
        self.row_off = 0
        self.has_precond = False

    def part_vector(self):
        """part[i] of every original row i, recovered from the permutation and rowPos."""
        part = np.empty(self.N, dtype=np.int32)
        part[self.perm] = np.repeat(np.arange(self.nparts, dtype=np.int32), np.diff(self.rowpos))
        return part

    # -- pieces of the driver --------------------------------------------------
    def create_block_jacobi(self):
        check(self.L.preAlps_BlockJacobiCreate(C.byref(self.A), self._rowpos_ptr, self._rowpos_n,
                                               self._colpos_ptr, self._colpos_n),
              "preAlps_BlockJacobiCreate")
        self.has_precond = True

    def reference_rhs(self):
        rhs = np.zeros(self.m)
        check(self.L.preAlps_hip_reference_rhs(_pd(rhs)), "preAlps_hip_reference_rhs")
        return rhs

    def local_csr(self):
        """Host copy of the local row panel (global column ids)."""
        m, nnz = self.A.info.m, self.A.info.lnnz
        rp = np.ctypeslib.as_array(self.A.rowPtr, shape=(m + 1,)).copy()
        ci = np.ctypeslib.as_array(self.A.colInd, shape=(max(nnz, 1),))[:nnz].copy()
        v = np.ctypeslib.as_array(self.A.val, shape=(max(nnz, 1),))[:nnz].copy()
        return rp, ci, v

    def stat(self, key):
        v = C.c_double()
        if self.L.preAlps_hip_get_stat(key.encode(), C.byref(v)) != 0:
            raise KeyError(key)
        return v.value

    def new_ecg(self, t, ortho_alg=ORTHODIR, bs_red=NO_BS_RED, tol=1e-5, max_iter=1000):
        e = preAlps_ECG_t()
        e.comm = 0x44000000
        e.globPbSize, e.locPbSize = self.M, self.m
        e.maxIter, e.enlFac, e.tol = max_iter, t, tol
        e.ortho_alg, e.bs_red = ortho_alg, bs_red
        return e

    def solve(self, rhs, t, ortho_alg=ORTHODIR, bs_red=NO_BS_RED, tol=1e-5, max_iter=1000):
        """preAlps_ECGSolve = the reference driver loop, in C."""
        if not self.has_precond:
            self.create_block_jacobi()
        import time
        L = self.L
        check(L.preAlps_hip_prepare_operator(int(t)), "preAlps_hip_prepare_operator")
        e = self.new_ecg(t, ortho_alg, bs_red, tol, max_iter)
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        sol = np.zeros(self.m)
        cap = max_iter + 2
        res = np.zeros(cap)
        bs = np.zeros(cap, dtype=np.int32)
        nh = C.c_int()
        t0 = time.perf_counter()
        check(L.preAlps_ECGSolve(C.byref(e), _pd(rhs), _pd(sol), _pd(res), _pi(bs), cap, C.byref(nh)),
              "preAlps_ECGSolve")
        dt = time.perf_counter() - t0
        timers = {k: getattr(e, k) for k in ("tot_t", "comm_t", "trsm_t", "gemm_t", "potrf_t", "copy_t")}
        return EcgResult(x=sol, iters=e.iter, res=res[:nh.value].copy(), bs=bs[:nh.value].copy(),
                         final_res=e.res, final_bs=e.bs, normb=e.normb, seconds=dt, timers=timers)

    # -- single operations, for tests and micro-benchmarks -----------------------
    def panel(self, ncols, t):
        """Allocate an m x ncols device panel laid out for enlarging factor t."""
        d = CPLM_Mat_Dense_t()
        check(self.L.preAlps_hip_panel_alloc(C.byref(d), self.M, ncols, self.m, ncols, t),
              "preAlps_hip_panel_alloc")
        return d

    def panel_free(self, d):
        self.L.preAlps_hip_panel_free(C.byref(d))

    def to_device(self, d, host, t):
        host = np.asfortranarray(host, dtype=np.float64)
        check(self.L.preAlps_hip_panel_from_host(C.byref(d), t, _pd(host), host.shape[0]),
              "preAlps_hip_panel_from_host")

    def to_host(self, d, t):
        out = np.zeros((d.info.m, d.info.n), order="F")
        check(self.L.preAlps_hip_panel_to_host(C.byref(d), t, _pd(out), max(d.info.m, 1)),
              "preAlps_hip_panel_to_host")
        return out

    def block_operator(self, X, t):
        """AX = A X for a host (m x n) array; goes through preAlps_BlockOperator."""
        n = X.shape[1]
        dx, dy = self.panel(n, t), self.panel(n, t)
        try:
            self.to_device(dx, X, t)
            check(self.L.preAlps_BlockOperator(C.byref(dx), C.byref(dy)), "preAlps_BlockOperator")
            return self.to_host(dy, t)
        finally:
            self.panel_free(dx)
            self.panel_free(dy)

    def block_jacobi_apply(self, X, t):
        if not self.has_precond:
            self.create_block_jacobi()
        n = X.shape[1]
        dx, dy = self.panel(n, t), self.panel(n, t)
        try:
            self.to_device(dx, X, t)
            check(self.L.preAlps_BlockJacobiApply(C.byref(dx), C.byref(dy)), "preAlps_BlockJacobiApply")
            return self.to_host(dy, t)
        finally:
            self.panel_free(dx)
            self.panel_free(dy)

    def sync(self):
        check(self.L.preAlps_hip_sync(), "preAlps_hip_sync")

    def close(self):
        self.L.preAlps_BlockJacobiFree()
        self.L.preAlps_OperatorFree()
        self.has_precond = False
