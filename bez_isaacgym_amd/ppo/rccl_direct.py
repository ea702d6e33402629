"""An RCCL communicator of the package's own, driven through RCCL's C API on the stream the caller names.

Why: `torch.distributed.all_reduce` runs the collective on the process group's internal stream and fences it against the caller's with two
cross-stream events.  In the data-parallel PPO update the gradient all-reduce sits between two HIP-graph replays on ONE stream, and those
two event waits are 14 us of idle GPU in front of each of an epoch's 20 optimiser launches (profiles/r05_epoch_timeline_dp.txt).  Issued with
`ncclAllReduce(..., stream)` on the training stream itself, segment / collective / segment are simply stream-ordered (VERDICT round 5,
next 4; the reference picks its device per rank and leaves the collectives to Horovod: utils/rlgames_utils.py:71-81).

The communicator is created next to torch's (same librccl the process already has loaded): rank 0 draws the unique id, it travels through the
existing process group, every rank calls ncclCommInitRank.  In-place fp32 / fp64 sums only -- all this package needs.  `available()` is
False without an initialised `nccl` (= RCCL) process group on a GPU, and the agent then keeps `torch.distributed` (gloo CPU tests).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
_NCCL_SUM = 0
_DTYPES = {torch.float32: 7, torch.float64: 8, torch.int32: 2, torch.int64: 4, torch.float16: 6}


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"]
        err = None
        for c in cands:
            try:
                _LIB = C.CDLL(c)
                break
            except OSError as e:   # noqa: PERF203
                err = e
        if _LIB is None:
            raise OSError("librccl not found (%s)" % err)
        _LIB.ncclGetUniqueId.restype = C.c_int; _LIB.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        _LIB.ncclCommInitRank.restype = C.c_int; _LIB.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        _LIB.ncclAllReduce.restype = C.c_int
        _LIB.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _LIB.ncclCommDestroy.restype = C.c_int; _LIB.ncclCommDestroy.argtypes = [C.c_void_p]
        _LIB.ncclGetErrorString.restype = C.c_char_p; _LIB.ncclGetErrorString.argtypes = [C.c_int]
    return _LIB


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, _lib().ncclGetErrorString(rc).decode(), rc))


def available(device):
    """an initialised RCCL process group and a GPU device: what a communicator of our own needs"""
    try:
        return (torch.device(device).type == "cuda" and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
                and os.environ.get("BEZ_PPO_DIRECT_RCCL", "1") != "0")
    except Exception:   # noqa: BLE001
        return False


class RcclComm:
    def __init__(self, device):
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        lib = _lib()
        uid = _UniqueId()
        if self.rank == 0:
            _chk(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        # the id travels through the process group that already exists (as a byte tensor on this rank's GPU: the nccl backend moves GPU tensors)
        # (C.string_at on the struct: reading the c_char array FIELD would stop at the id's first zero byte)
        t = torch.frombuffer(bytearray(C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES) if self.rank == 0 else bytes(NCCL_UNIQUE_ID_BYTES)), dtype=torch.uint8).to(self.device)
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().numpy().tobytes())
        C.memmove(C.addressof(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _chk(lib.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.calls = 0

    def all_reduce_(self, t, stream=None):
        """in-place sum over the ranks, enqueued on `stream` (default: torch's current stream on this device): no event, no side stream"""
        assert t.is_cuda and t.is_contiguous() and t.dtype in _DTYPES, (t.device, t.dtype)
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        _chk(_lib().ncclAllReduce(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), _DTYPES[t.dtype], _NCCL_SUM, self._comm, C.c_void_p(st.cuda_stream)),
             "ncclAllReduce")
        self.calls += 1
        return t

    def close(self):
        if getattr(self, "_comm", None) is not None and self._comm.value:
            try:
                _lib().ncclCommDestroy(self._comm)
            except Exception:   # noqa: BLE001
                pass
            self._comm = C.c_void_p()

    def __del__(self):
        self.close()
