"""Launcher glue for more than one rank (one process per GPU, started by torch.distributed.run).

torch.distributed (gloo) is plumbing only: it carries the host-side set-up collectives the reference
does with MPI (layout all-gather, ghost-column lists) and distributes the RCCL unique id.  The data
path -- halo exchange and scalar all-reduces on device buffers -- is RCCL over xGMI called from C
(include/mi355x_comm.h), never through Python."""
import ctypes as C
import os
import sys

import numpy as np

from . import petsc as P

_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
_AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int)
_BAR = C.CFUNCTYPE(C.c_int, C.c_void_p)
_EXC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                   C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int))
_keep = []


def make_comm(rank, size, allgather_bytes, allreduce_array, barrier, exchange=None):
    """Build a PetscComm from Python callables:
       allgather_bytes(bytes) -> list of `size` bytes objects; allreduce_array(np.ndarray, op) -> np.ndarray; barrier();
       exchange(sends=[(peer, bytes)], recvs=[(peer, nbytes)]) -> list of bytes, one per recv (host-staged halo, optional)."""
    L = P.lib()

    def ag(ctx, sendbuf, nbytes, recvbuf):
        try:
            parts = allgather_bytes(C.string_at(sendbuf, nbytes))
            C.memmove(recvbuf, b"".join(parts), nbytes * size)
            return 0
        except Exception as e:  # pragma: no cover
            print("allgather callback failed:", e, flush=True)
            return 1

    def ar(ctx, buf, count, is_double, op):
        try:
            dt = np.float64 if is_double else np.int32
            a = np.frombuffer(C.string_at(buf, count * np.dtype(dt).itemsize), dtype=dt).copy()
            r = np.ascontiguousarray(allreduce_array(a, op), dtype=dt)
            C.memmove(buf, r.ctypes.data, r.nbytes)
            return 0
        except Exception as e:  # pragma: no cover
            print("allreduce callback failed:", e, flush=True)
            return 1

    def bar(ctx):
        barrier()
        return 0

    def exc(ctx, ns, speers, sbufs, sbytes, nr, rpeers, rbufs, rbytes):
        try:
            sends = [(speers[i], C.string_at(sbufs[i], sbytes[i])) for i in range(ns)]
            recvs = [(rpeers[i], rbytes[i]) for i in range(nr)]
            got = exchange(sends, recvs)
            for i in range(nr):
                C.memmove(rbufs[i], got[i], rbytes[i])
            return 0
        except Exception as e:  # pragma: no cover
            print("exchange callback failed:", e, flush=True)
            return 1

    cbs = (_AG(ag), _AR(ar), _BAR(bar), _EXC(exc))
    _keep.append(cbs)
    comm = C.c_void_p()
    L.PetscCommCreate(rank, size, None, C.cast(cbs[0], C.c_void_p), C.cast(cbs[1], C.c_void_p), C.cast(cbs[2], C.c_void_p), C.byref(comm))
    if exchange is not None:
        L.PetscCommSetExchange(comm, C.cast(cbs[3], C.c_void_p))
    return comm


last_transport = "single"   # what the last torch_comm() call ended up with: "single", "rccl" or "host-staged"


def _rccl_self_test(k, dcomm, rank, size):
    """One all-reduce and one ring send/recv on a fresh communicator, checked on the host; returns "" or what failed."""
    h = C.c_void_p()
    if k.mi355x_handle_create(C.byref(h)):
        return "handle"
    buf = C.c_void_p()
    try:
        if k.mi355x_malloc(C.byref(buf), 64):
            return "malloc"
        v = np.array([1.0 + rank, -1.0, 0.0, 0.0])
        k.mi355x_memcpy_h2d(h, buf, v.ctypes.data, v.nbytes)
        rc = k.mi355x_comm_allreduce_sum(dcomm, h, buf, 1)
        if rc:
            return "ncclAllReduce: %s" % k.mi355x_comm_error_string(rc).decode()
        nxt, prv = (rank + 1) % size, (rank - 1) % size
        rc = k.mi355x_comm_group_start()
        rc = rc or k.mi355x_comm_recv(dcomm, h, C.c_void_p(buf.value + 16), 1, prv)
        rc = rc or k.mi355x_comm_send(dcomm, h, C.c_void_p(buf.value + 8), 1, nxt)
        rc2 = k.mi355x_comm_group_end()
        if rc or rc2:
            return "ncclSend/Recv: %s" % k.mi355x_comm_error_string(rc or rc2).decode()
        k.mi355x_memcpy_d2h(h, v.ctypes.data, buf, v.nbytes)
        if k.mi355x_handle_synchronize(h):
            return "stream synchronise after the self test"
        if v[0] != size * (size + 1) / 2.0 or v[2] != -1.0:
            return "self test values %r" % (v.tolist(),)
        return ""
    finally:
        if buf:
            k.mi355x_free(buf)
        k.mi355x_handle_destroy(h)


def torch_comm(device_comm=True):
    """PetscComm over the default torch.distributed (gloo) group; optionally attaches an RCCL communicator."""
    import torch
    import torch.distributed as dist
    rank, size = dist.get_rank(), dist.get_world_size()
    ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}

    def allgather_bytes(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        outs = [torch.empty_like(t) for _ in range(size)]
        dist.all_gather(outs, t)
        return [bytes(o.numpy().tobytes()) for o in outs]

    def allreduce_array(a, op):
        t = torch.from_numpy(a.astype(np.float64 if a.dtype == np.float64 else np.int64))
        dist.all_reduce(t, op=ops[op])
        return t.numpy().astype(a.dtype)

    def exchange(sends, recvs):
        bufs = [torch.empty(nb, dtype=torch.uint8) for _, nb in recvs]
        ops = [dist.P2POp(dist.irecv, b, peer) for b, (peer, _) in zip(bufs, recvs)]
        keep = [torch.frombuffer(bytearray(data), dtype=torch.uint8) for _, data in sends]
        ops += [dist.P2POp(dist.isend, t, peer) for t, (peer, _) in zip(keep, sends)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return [bytes(b.numpy().tobytes()) for b in bufs]

    comm = make_comm(rank, size, allgather_bytes, allreduce_array, dist.barrier, exchange=exchange)
    L = P.lib()
    L.PetscCommSetWorld(comm)
    if device_comm and size > 1:
        k = P.load_kernels()
        ndev = C.c_int()
        k.mi355x_device_count(C.byref(ndev))
        dev = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev.value, 1)
        dcomms, why = [], ""
        rc = k.mi355x_set_device(dev)
        if rc:   # no exception here: the other ranks are about to enter a broadcast and must not be left waiting
            why = "hipSetDevice(%d) failed: %s" % (dev, k.mi355x_error_string(rc).decode())
        # TWO communicators over the same ranks, one per HIP stream: RCCL runs the operations of one communicator one
        # after the other in issue order whatever streams they are given, so the halo exchange (halo stream) and the
        # scalar all-reduces (compute stream) each get their own.  Every rank issues its operations in the same
        # program order (the solvers are SPMD), which is what concurrent use of two communicators requires.
        for which in ("reductions", "halo"):
            uid = C.create_string_buffer(128)
            have_id = True
            if rank == 0:
                rc = k.mi355x_comm_get_unique_id(uid) if not why else 1
                if rc:
                    have_id = False
                    why = why or "ncclGetUniqueId failed: %s" % k.mi355x_comm_error_string(rc).decode()
            obj = [uid.raw if have_id else b""]
            dist.broadcast_object_list(obj, src=0)
            if not obj[0]:
                why = why or "rank 0 could not create an RCCL unique id"
            ok = torch.tensor([0 if why else 1], dtype=torch.int32)      # everybody continues, or nobody
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) != 1:
                why = why or "another rank failed before ncclCommInitRank"
                break
            dcomm = C.c_void_p()
            rc = k.mi355x_comm_init_rank(C.byref(dcomm), size, rank, obj[0])
            if rc:
                why = why or "ncclCommInitRank (%s): %s" % (which, k.mi355x_comm_error_string(rc).decode())
            else:
                dcomms.append(dcomm)
            # every rank must take the same transport, and no rank may enter a collective of a communicator another rank
            # failed to join: agree over gloo after the initialisation, then again after the self test
            ok = torch.tensor([0 if why else 1], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) != 1:
                why = why or "another rank failed in ncclCommInitRank"
                break
            why = _rccl_self_test(k, dcomm, rank, size)
            ok = torch.tensor([0 if why else 1], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) != 1:
                why = why or "another rank failed the RCCL self test"
                break
        if not why:
            L.PetscCommSetDeviceComms(comm, dcomms[0], dcomms[1])
            transport = "rccl"
        else:
            # LOUD: the halo and the reductions then travel device -> host -> gloo -> host -> device (the reference's own
            # CUSP arrangement); still the GPU compute path, but not the transport this library is built for
            print("[petsc-hipmi355x] rank %d: RCCL communicator unusable (%s); ALL ranks fall back to the host-staged transport"
                  % (rank, why), file=sys.stderr, flush=True)
            for d in dcomms:
                k.mi355x_comm_destroy(d)
            transport = "host-staged"
    elif size > 1:
        transport = "host-staged"
    else:
        transport = "single"
    global last_transport
    last_transport = transport
    return comm


def transport_report(comm):
    """What the device-side collectives of `comm` really travel over, asked of the C library (not of this module's
    bookkeeping): {"transport": "single"|"rccl"|"host-staged", "rccl_ranks": n RCCL reports, "rccl_communicators": 1|2}"""
    L = P.lib()
    kind, nranks, distinct = C.c_int(), C.c_int(), C.c_int()
    L.PetscCommGetDeviceTransport(comm, C.byref(kind), C.byref(nranks), C.byref(distinct))
    return {"transport": ("single", "rccl", "host-staged")[kind.value], "rccl_ranks": nranks.value,
            "rccl_communicators": (2 if distinct.value else 1) if kind.value == 1 else 0}
