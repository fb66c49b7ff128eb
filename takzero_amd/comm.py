"""The exchange between the self-play shards of one job over the C ABI (tz_comm_*, csrc/tz_comm.cpp):
RCCL on the shard's GPU (ncclAllGather / ncclBroadcast over xGMI inside a node) or files of a shared directory ("fs").
The reference has no collective: its N processes append to the same files of one directory (README.md:130)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

ID_BYTES = 128


def unique_id():
    """ncclGetUniqueId (one rank calls it and carries the bytes to the others)."""
    buf = (C.c_ubyte * ID_BYTES)()
    check(_lib.load().tz_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    def __init__(self, handle, kind):
        self.lib, self.h, self.kind = _lib.load(), handle, kind

    @classmethod
    def rccl(cls, comm_id, rank, world, device):
        h = C.c_void_p()
        buf = (C.c_ubyte * ID_BYTES).from_buffer_copy(comm_id)
        check(_lib.load().tz_comm_create_rccl(buf, rank, world, device, C.byref(h)))
        return cls(h, "rccl")

    @classmethod
    def rccl_from_directory(cls, directory, rank, world, device, timeout_s=120.0):
        """Rank 0 publishes the id as `<directory>/rccl_id.bin`, the others wait for it (a fresh directory per job)."""
        buf = (C.c_ubyte * ID_BYTES)()
        check(_lib.load().tz_comm_rendezvous_id(str(directory).encode(), rank, buf, timeout_s))
        return cls.rccl(bytes(buf), rank, world, device)

    @classmethod
    def fs(cls, directory, rank, world, timeout_s=600.0):
        h = C.c_void_p()
        check(_lib.load().tz_comm_create_fs(str(directory).encode(), rank, world, timeout_s, C.byref(h)))
        return cls(h, "fs")

    def close(self):
        if getattr(self, "h", None):
            self.lib.tz_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        r, w, k = C.c_int(), C.c_int(), C.c_int()
        n, b = C.c_uint64(), C.c_uint64()
        check(self.lib.tz_comm_info(self.h, C.byref(r), C.byref(w), C.byref(k), C.byref(n), C.byref(b)))
        return dict(rank=r.value, world=w.value, transport="rccl" if k.value else "fs", collectives=n.value, bytes_gathered=b.value)

    def all_gather(self, data):
        """bytes of every rank, as a list in rank order."""
        data = bytes(data)
        world = self.info()["world"]
        sizes = np.zeros(world, np.uint64)
        total = C.c_uint64()
        src = (C.c_char * max(1, len(data))).from_buffer_copy(data or b"\0")
        check(self.lib.tz_comm_all_gather(self.h, src, len(data), sizes.ctypes.data, C.byref(total)))
        out = (C.c_char * max(1, int(total.value)))()
        check(self.lib.tz_comm_take(self.h, out, len(out)))
        raw, parts, at = bytes(out[:total.value]), [], 0
        for s in sizes:
            parts.append(raw[at:at + int(s)])
            at += int(s)
        return parts

    def broadcast(self, data, root=0):
        buf = (C.c_char * len(data)).from_buffer_copy(bytes(data))
        check(self.lib.tz_comm_broadcast(self.h, buf, len(data), root))
        return bytes(buf)

    def barrier(self):
        check(self.lib.tz_comm_barrier(self.h))

    def broadcast_net(self, net, root=0, status=0):
        """tz_net_broadcast: `status` = the root's own load result (0 = a new model is active there)."""
        check(self.lib.tz_net_broadcast(net.h, self.h, root, status))
