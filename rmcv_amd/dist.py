"""Multi-GPU: a batch of independent camera frames sharded over the ranks of one node.

The reference is single-process (one process_thread, executable/main.cpp:55); frames carry no
cross-frame state on this path, so rank r simply owns the contiguous block of frames
[r*n/R, (r+1)*n/R).  There is no collective on the data path; the only exchange is the final gather
of the (tiny, variable-length) armour lists to rank 0:

    ONE torch.distributed.gather of a fixed-size record per rank
        [ frame_offs : n_local+1 int32, padded to 16 B | armours : cap x 88 bytes ]
    (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)

The device-side compaction kernel (rmcv_batch_compact_armours) writes straight into the record, so a
step is: detect -> compact -> gather, with no host round trip.  A gather lowers to grouped send/recv,
i.e. every peer uses its own direct xGMI link to the root: the payload is O(100 KB), latency-bound,
and must not be pushed round a ring.  frame_offs[-1] is the rank's true armour count; a count above
`cap` is detected by unpack_records (the caller then re-gathers with a larger cap).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

ARMOUR_BYTES = 88


def shard(n_total, rank, world):
    """contiguous block of frame indices owned by `rank`"""
    return n_total * rank // world, n_total * (rank + 1) // world


def record_layout(n_local, cap):
    """(offset of armours, total bytes) of a rank's record: [frame_offs: n_local + 1 int32 | status: int32 | dense frames: int32 |
    pad to 16 B | armours] -- the record rmcv_pipeline_* keeps per batch (include/rmcv_abi.h: rmcv_pipeline_info): the status word
    is the OR of the batch's per-frame status bits, the next one counts its frames beyond findContours' LDS tables"""
    head = ((n_local + 3) * 4 + 15) // 16 * 16
    return head, head + cap * ARMOUR_BYTES


def new_record(n_local, cap, device):
    _, total = record_layout(n_local, cap)
    return torch.zeros(total, dtype=torch.uint8, device=device)


def fill_record(rec, n_local, cap, frame_offs, armours_u8):
    """host-side/packaged fill (tests, CPU): frame_offs int32[n_local+1], armours uint8[total*88]"""
    head, _ = record_layout(n_local, cap)
    rec[:(n_local + 1) * 4] = torch.as_tensor(np.ascontiguousarray(frame_offs, np.int32).view(np.uint8)).to(rec.device)
    a = torch.as_tensor(np.ascontiguousarray(armours_u8, np.uint8).reshape(-1)).to(rec.device)
    k = min(a.numel(), cap * ARMOUR_BYTES)
    rec[head:head + k] = a[:k]
    return rec


def new_gather_list(rec, group=None, dst=0):
    """receive buffers for gather_records on dst (None elsewhere), to be allocated once and reused every step"""
    if not dist.is_initialized():
        return None
    return [torch.empty_like(rec) for _ in range(dist.get_world_size(group))] if dist.get_rank(group) == dst else None


def gather_records(rec, group=None, dst=0, out=None, async_op=False):
    """the one collective of the path; returns the list of records on dst, None elsewhere.  `out`: a list from new_gather_list.
    async_op=True returns (records, work): the collective is ordered after the work already enqueued on the current stream but the
    current stream does NOT wait for it -- a pipelined caller goes on with the next step's kernels and calls work.wait() (a
    stream-side wait, no host block) before it rewrites `rec` or reads the records."""
    if not dist.is_initialized():
        return ([rec], None) if async_op else [rec]
    rank = dist.get_rank(group)
    recs = (out if out is not None else [torch.empty_like(rec) for _ in range(dist.get_world_size(group))]) if rank == dst else None
    work = dist.gather(rec, recs, dst=dst, group=group, async_op=async_op)
    return (recs, work) if async_op else recs


def unpack_records(recs, n_local, cap):
    """records (rank order) -> (armours uint8 [total, 88], global frame_offs int64 [R*n_local+1]).
    Raises OverflowError if some rank had more than `cap` armours."""
    head, _ = record_layout(n_local, cap)
    arms, offs, base = [], [0], 0
    for r, rec in enumerate(recs):
        h = rec.detach().cpu().numpy()
        fo = h[:(n_local + 1) * 4].view(np.int32).astype(np.int64)
        tot = int(fo[-1])
        if tot > cap:
            raise OverflowError("rank %d produced %d armours > record capacity %d" % (r, tot, cap))
        arms.append(h[head:head + tot * ARMOUR_BYTES].reshape(tot, ARMOUR_BYTES))
        offs.extend((fo[1:] + base).tolist())
        base += tot
    arm = np.concatenate(arms, axis=0) if arms else np.zeros((0, ARMOUR_BYTES), np.uint8)
    return arm, np.asarray(offs, np.int64)


def gather_detections(frame_offs, armours_u8, cap, device="cpu", group=None, dst=0):
    """convenience form for host-side lists: pack, gather, unpack (on dst; None elsewhere)"""
    n_local = len(frame_offs) - 1
    rec = fill_record(new_record(n_local, cap, device), n_local, cap, frame_offs, armours_u8)
    recs = gather_records(rec, group, dst)
    if recs is None:
        return None
    return unpack_records(recs, n_local, cap)


class _DevMem:
    """`nbytes` bytes at a device address, for torch.as_tensor (a view, no copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def tensor_at(ptr, nbytes, device):
    """uint8 tensor VIEW of `nbytes` bytes at `ptr` -- device memory for a cuda device, host memory for "cpu" (the gloo tests)"""
    device = torch.device(device)
    if device.type == "cuda":
        return torch.as_tensor(_DevMem(ptr, nbytes), device=device)
    return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(int(ptr))))


class TorchGatherHook:
    """The hook of a Pipeline (rmcv_pipeline_set_hook) that gathers every batch's record to `dst` with torch.distributed --
    RCCL over xGMI on the GPU box (backend "nccl"), gloo in the CPU tests.  Asynchronous: the collective is ordered behind the
    record's compaction (the stream the pipeline names), the pipeline's streams do not wait for it; a record is rewritten `depth`
    tickets later, and the pipeline orders that rewrite behind the event this hook hands back (cuda) -- a CPU caller waits with
    wait(ticket) before it refills the record."""

    def __init__(self, record_bytes, depth, device, group=None, dst=0):
        self.depth, self.device, self.group, self.dst = int(depth), torch.device(device), group, dst
        self.cuda = self.device.type == "cuda"
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.record_bytes = int(record_bytes)
        self.recs = [None] * self.depth
        self.outs = [[torch.empty(self.record_bytes, dtype=torch.uint8, device=self.device) for _ in range(self.world)]
                     if self.rank == dst else None for _ in range(self.depth)]
        self.works = [None] * self.depth
        self._ext = {}
        if self.cuda:
            self.side = torch.cuda.Stream(device=self.device)
            self.events = [torch.cuda.Event() for _ in range(self.depth)]

    def __call__(self, ticket, d_record, record_bytes, hip_stream):
        k = ticket % self.depth
        if self.recs[k] is None or self.recs[k].data_ptr() != d_record:
            self.recs[k] = tensor_at(d_record, record_bytes, self.device)
        if not self.cuda:
            self.works[k] = dist.gather(self.recs[k], self.outs[k], dst=self.dst, group=self.group, async_op=True)
            return None
        ext = self._ext.get(hip_stream)
        if ext is None:
            ext = self._ext[hip_stream] = torch.cuda.ExternalStream(hip_stream, device=self.device)
        with torch.cuda.stream(ext):        # the process group's stream waits for this one: the record is complete when it is read
            self.works[k] = dist.gather(self.recs[k], self.outs[k], dst=self.dst, group=self.group, async_op=True)
        with torch.cuda.stream(self.side):  # ... and the record's next rewrite waits for the collective, not the other way round
            self.works[k].wait()
            self.events[k].record(self.side)
        return self.events[k].cuda_event

    def wait(self, ticket):
        """host-side: the gather of `ticket` is through (its record may be rewritten, its gathered records read)"""
        k = ticket % self.depth
        if self.works[k] is not None:
            self.works[k].wait()
            if self.cuda:
                self.events[k].synchronize()

    def wait_all(self):
        for k in range(self.depth):
            if self.works[k] is not None:
                self.works[k].wait()
        if self.cuda:
            torch.cuda.synchronize(self.device)

    def records(self, ticket):
        """dst: the list of per-rank records of `ticket` (valid after wait, until ticket + depth is submitted); None elsewhere"""
        return self.outs[ticket % self.depth]


class AbiGather:
    """The same gather through the C-ABI (rmcv_comm_* / rmcv_gather in librmcv_hip.so): RCCL called by the library itself, which
    is what a C++ host uses (include/rmcv_abi.h).  torch.distributed only carries the 128-byte group id from rank 0 to the
    others here; the payload moves by ncclSend/ncclRecv on the caller's HIP stream."""

    def __init__(self, device_index, group=None):
        from .abi import COMM_ID_BYTES, RmcvError, lib
        self._lib, self._err = lib(), RmcvError
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        idt = torch.zeros(COMM_ID_BYTES, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_uint8 * COMM_ID_BYTES)()
            self._chk(self._lib.rmcv_comm_unique_id(buf), None)
            idt = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
        dev = torch.device("cuda", device_index) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        idt = idt.to(dev)
        dist.broadcast(idt, src=0, group=group)
        raw = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(idt.cpu().numpy()))
        h = C.c_void_p()
        self._chk(self._lib.rmcv_comm_create(raw, self.world, self.rank, int(device_index), C.byref(h)), None)
        self._h = h
        n, r = C.c_int32(0), C.c_int32(0)
        self._chk(self._lib.rmcv_comm_info(self._h, C.byref(n), C.byref(r)), self._h)
        assert (n.value, r.value) == (self.world, self.rank)

    def _chk(self, rc, h):
        if rc != 0:
            self._lib.rmcv_comm_last_error.restype = C.c_char_p
            self._lib.rmcv_comm_last_error.argtypes = [C.c_void_p]
            raise self._err(rc, self._lib.rmcv_comm_last_error(h).decode())

    def new_recv(self, rec):
        """root's receive buffer (world x record bytes, one tensor), None elsewhere"""
        return torch.empty(self.world * rec.numel(), dtype=torch.uint8, device=rec.device) if self.rank == 0 else None

    def gather(self, rec, recv, stream=None):
        """enqueue on `stream` (a hipStream_t as int; None = the null stream); returns the list of per-rank records (views) on root"""
        self._chk(self._lib.rmcv_gather(self._h, C.c_void_p(rec.data_ptr()), C.c_int64(rec.numel()),
                                        C.c_void_p(recv.data_ptr() if recv is not None else 0), 0, C.c_void_p(stream or 0)), self._h)
        if recv is None:
            return None
        n = rec.numel()
        return [recv[r * n:(r + 1) * n] for r in range(self.world)]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rmcv_comm_destroy.restype = None
            self._lib.rmcv_comm_destroy.argtypes = [C.c_void_p]
            self._lib.rmcv_comm_destroy(self._h)
            self._h = None
