"""Exchange hooks for svt_set_shard (include/savont_hip.h): the in-place all-gather-v the device layer calls when one rank has run only its
slice of a call's tiles (K3/K4 by read block, K5/K6 tile rows by block, all-gather of what the other ranks need).

  TorchExchange   one process per GPU, torch.distributed (RCCL on the GPU box; gloo on host memory in CPU tests): one broadcast per rank over
                  views of the SAME device array -- no staging, no host hop
  LocalExchange   `world` pipelines of ONE process on ONE device, each driven by its own thread (tests: the rank logic of the library on a
                  single-GPU box): the ranks meet at a barrier and copy their slices device-to-device

No torch import at module level: pipeline.load() must stay torch-free."""
import ctypes as C
import threading

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64))


class _DevMem:
    """a raw device range as a __cuda_array_interface__ object (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchExchange:
    """group: the process group to broadcast on (default: the world).  stage_host: the group's backend cannot move device memory (gloo) while the
    arrays live on a GPU -- the oversubscribed test mode of bench.py, several ranks on ONE device: every slice hops through host memory"""

    def __init__(self, dist, device, world, rank, group=None, stage_host=False, fail_at=None):
        import torch
        self.fail_at = fail_at                                # test hook: this exchange (counted from 0) fails on this rank without taking part in it
        self.torch = torch; self.dist = dist; self.device = device; self.world = world; self.rank = rank
        self.group = group; self.stage_host = bool(stage_host) and device.type == "cuda"
        self.calls = 0; self.bytes = 0
        self.hook = EXCHANGE_FN(self._call)                   # keep the object alive as long as the library may call it

    def _view(self, ptr, nbytes):
        torch = self.torch
        if self.device.type == "cuda":
            return torch.as_tensor(_DevMem(ptr, nbytes), device=self.device)
        return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)

    def _call(self, user, base, elem_bytes, off):
        try:
            o = [int(off[r]) * int(elem_bytes) for r in range(self.world + 1)]
            if o[-1] == o[0]:
                return 0
            if self.fail_at is not None and self.calls == self.fail_at:
                import sys
                print("shard exchange %d fails on rank %d by request (test hook)" % (self.calls, self.rank), file=sys.stderr)
                return 1
            t = self._view(int(base) + o[0], o[-1] - o[0])
            for r in range(self.world):
                if o[r + 1] > o[r]:
                    sl = t[o[r] - o[0]:o[r + 1] - o[0]]
                    if self.stage_host:
                        h = sl.cpu() if r == self.rank else self.torch.empty(sl.numel(), dtype=self.torch.uint8)
                        self.dist.broadcast(h, src=r, group=self.group)
                        if r != self.rank:
                            sl.copy_(h)
                    else:
                        self.dist.broadcast(sl, src=r, group=self.group)
            if self.device.type == "cuda":
                self.torch.cuda.synchronize(self.device)
            self.calls += 1; self.bytes += o[-1] - o[0]
            return 0
        except Exception as e:                                    # an exception must not unwind through the C caller
            import sys
            print("shard exchange failed on rank %d: %r" % (self.rank, e), file=sys.stderr)
            return 1


class LocalExchange:
    """world ranks = world threads of this process on one device; rank r's hook copies its slice into every other rank's array"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.base = [0] * world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemcpy.restype = C.c_int
        self.hooks = [EXCHANGE_FN(self._make(r)) for r in range(world)]
        self.calls = 0; self.bytes = 0; self.failed = False

    def _make(self, rank):
        def call(user, base, elem_bytes, off):
            try:
                o = [int(off[r]) * int(elem_bytes) for r in range(self.world + 1)]
                self.base[rank] = int(base)
                self.barrier.wait(timeout=120)                    # every rank has published its array (and synchronised its stream)
                rc = 0
                if o[rank + 1] > o[rank]:
                    for q in range(self.world):
                        if q != rank:
                            rc |= self.hip.hipMemcpy(self.base[q] + o[rank], int(base) + o[rank], o[rank + 1] - o[rank], 3)   # hipMemcpyDeviceToDevice
                if rank == 0:
                    self.calls += 1; self.bytes += o[-1] - o[0]
                self.barrier.wait(timeout=120)                    # every slice has arrived everywhere
                return 0 if rc == 0 else 1
            except Exception as e:
                import sys
                self.failed = True
                print("local shard exchange failed on rank %d: %r" % (rank, e), file=sys.stderr)
                try:
                    self.barrier.abort()
                except Exception:
                    pass
                return 1
        return call
