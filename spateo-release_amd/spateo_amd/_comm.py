"""The EM step's exchange through the C ABI: ``mvf_comm_*`` / ``mvf_allreduce_stats`` (include/mvf.h) = RCCL over xGMI.

``MvfComm`` owns one ``mvf_comm`` handle (an RCCL communicator bound to one GPU).  The 128-byte unique id is created by
rank 0 and handed to the other ranks out of band: through ``torch.distributed`` when a process group exists (any backend:
it only moves 128 bytes once), directly when there is one rank.  After that the data path does not touch torch's
collectives: ``all_reduce`` enqueues ``ncclAllReduce`` on the stream it is given, in place on a float64 device tensor.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class MvfComm:
    def __init__(self, device, rank=0, world=1, group=None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.rank, self.world = int(rank), int(world)
        idbuf = C.create_string_buffer(_lib.MVF_COMM_ID_BYTES)
        if self.rank == 0:
            _lib.check(self.lib.mvf_comm_unique_id(idbuf), "mvf_comm_unique_id")
        if self.world > 1:
            import torch.distributed as dist

            box = [bytes(idbuf.raw) if self.rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            with torch.cuda.device(self.device):
                dist.broadcast_object_list(box, src=src, group=group)
            idbuf = C.create_string_buffer(box[0], _lib.MVF_COMM_ID_BYTES)
        handle = C.c_void_p(None)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mvf_comm_create(C.byref(handle), self.world, self.rank, idbuf), "mvf_comm_create")
        self._h = handle

    def info(self):
        """(nranks, rank, device index) as the library reports them."""
        n, r, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        _lib.check(self.lib.mvf_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d)), "mvf_comm_info")
        return n.value, r.value, d.value

    def all_reduce(self, t, op="sum", stream=None):
        """In place on the float64 device tensor `t`, asynchronous on `stream` (default: torch's current stream)."""
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("MvfComm.all_reduce needs a contiguous float64 device tensor")
        if self._h is None:
            raise RuntimeError("MvfComm.all_reduce after close()")
        if op not in ("sum", "min"):
            raise ValueError(f"MvfComm.all_reduce: op must be 'sum' or 'min', not {op!r}")
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mvf_allreduce_stats(self._h, t.data_ptr(), t.numel(),
                                                    _lib.RED_SUM if op == "sum" else _lib.RED_MIN, s.cuda_stream),
                       "mvf_allreduce_stats")

    def close(self):
        h, self._h = self._h, None
        if h is not None:
            torch.cuda.synchronize(self.device)
            _lib.check(self.lib.mvf_comm_destroy(h), "mvf_comm_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
