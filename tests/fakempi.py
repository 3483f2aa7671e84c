"""An in-process "fake MPI" for CPU tests (the role MPIUNI plays in the reference's test strategy):
every rank is a Python thread; the three host collectives rendezvous on a threading.Barrier."""
import threading

import numpy as np

from petsc_dev_amd import dist as D


class FakeWorld:
    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size
        self.boxes = {}

    def comm_for(self, rank):
        def allgather_bytes(b):
            self.slots[rank] = b
            self.barrier.wait()
            out = list(self.slots)
            self.barrier.wait()
            return out

        def allreduce_array(a, op):
            parts = allgather_bytes(a.tobytes())
            arrs = np.stack([np.frombuffer(p, dtype=a.dtype) for p in parts])
            return {0: arrs.sum(0), 1: arrs.max(0), 2: arrs.min(0)}[op].astype(a.dtype)

        def exchange(sends, recvs):
            import queue
            for peer, data in sends:
                self.boxes.setdefault((rank, peer), queue.Queue()).put(data)
            out = []
            for peer, nb in recvs:
                while True:
                    q = self.boxes.get((peer, rank))
                    if q is not None:
                        break
                    import time
                    time.sleep(0.0005)
                out.append(q.get(timeout=60))
            return out

        return D.make_comm(rank, self.size, allgather_bytes, allreduce_array, lambda: self.barrier.wait(), exchange=exchange)

    def run(self, fn):
        """fn(rank, comm) on every rank; returns the list of results; re-raises the first failure"""
        res = [None] * self.size
        err = [None] * self.size

        def work(r):
            try:
                res[r] = fn(r, self.comm_for(r))
            except BaseException as e:  # noqa
                err[r] = e
                self.barrier.abort()
        th = [threading.Thread(target=work, args=(r,)) for r in range(self.size)]
        [t.start() for t in th]
        [t.join() for t in th]
        for e in err:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return res
