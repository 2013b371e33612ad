"""Rendezvous of the ranks of a one-node multi-GPU job WITHOUT torch in the rank processes.

Why not torch.distributed: torch's wheel bundles its own HIP runtime, HSA runtime and RCCL (torch/lib/libamdhip64.so, SONAME
libamdhip64.so.7 - the SONAME of the system library libugs.so links).  A process that imports torch FIRST runs the product's kernels on
torch's ROCm 7.0 runtime and the product's gather on torch's RCCL; a process that loads libugs.so first and torch second maps BOTH
runtimes (torch asks for "libamdhip64.so", which no loaded SONAME satisfies) and dies of a corrupted heap at exit.  The ranks of
`bench.py --gpus N` therefore never import torch: one HIP runtime, one RCCL - the system ones - by construction.

What the ranks need from each other is tiny (the RCCL unique id, a barrier, the maximum of a timing, a few statistics): a star of TCP
connections on 127.0.0.1 to rank 0, found through a file in /dev/shm named after MASTER_PORT (which holds rank 0's own port and pid).  The launcher itself may be torch.distributed.run (the driver's command line) - it only exports RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT and is a process of its own.

`GlooGroup` offers the same interface over an initialised torch.distributed process group (the world-2 gloo tests on CPU, which run
in processes that never load libugs.so)."""
import os
import pickle
import socket
import struct
import tempfile
import time


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)))
    sock.sendall(data)


def _recv(sock):
    def exact(n):
        buf = bytearray(n)
        view, got = memoryview(buf), 0
        while got < n:
            k = sock.recv_into(view[got:], n - got)
            if k == 0:
                raise ConnectionError("a rank of the job went away")
            got += k
        return buf
    (n,) = struct.unpack("<Q", exact(8))
    return pickle.loads(exact(n))


def _rdzv_path(tag):
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    return os.path.join(shm, "ugs_hostgroup_%s" % tag)


class SocketGroup:
    """allgather / broadcast / barrier / max over the ranks of one node; rank 0 is the hub."""

    def __init__(self, rank, world, tag=None, timeout=600.0):
        self.rank, self.world = int(rank), int(world)
        self.peers, self.sock, self.path = [], None, None
        if self.world <= 1:
            return
        if tag is None:                                          # MASTER_PORT is unique to a job on this node while the job lives
            tag = "port%s" % os.environ.get("MASTER_PORT", "0")
        self.path = _rdzv_path(tag)
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            tmp = self.path + ".tmp%d" % os.getpid()
            with open(tmp, "w") as f:
                f.write("%d %d\n" % (srv.getsockname()[1], os.getpid()))
            os.replace(tmp, self.path)                           # (atomic: a reader sees the whole line or the old file)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                r = _recv(c)
                by_rank[int(r)] = c
            srv.close()
            self.peers = [by_rank[r] for r in range(1, self.world)]
            try:
                os.remove(self.path)
            except OSError:
                pass
        else:
            t0 = time.time()
            while True:
                port = self._read_port()
                if port:
                    try:
                        s = socket.create_connection(("127.0.0.1", port), timeout=5.0)
                        break
                    except OSError:
                        pass
                if time.time() - t0 > timeout:
                    raise TimeoutError("rank %d: no rendezvous file %s from rank 0" % (self.rank, self.path))
                time.sleep(0.02)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            _send(s, self.rank)
            self.sock = s

    def _read_port(self):
        """the hub's port from the rendezvous file - only if the process that wrote it is alive (a stale file of an earlier job is ignored)"""
        try:
            port, pid = (int(x) for x in open(self.path).read().split())
        except (OSError, ValueError):
            return 0
        return port if os.path.exists("/proc/%d" % pid) else 0

    def allgather(self, obj):
        if self.world <= 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [_recv(c) for c in self.peers]
            for c in self.peers:
                _send(c, out)
            return out
        _send(self.sock, obj)
        return _recv(self.sock)

    def gather(self, obj, dst=0):
        """`obj` of every rank on rank `dst` (a list in rank order), None elsewhere; large payloads travel once"""
        if self.world <= 1:
            return [obj]
        if dst != 0:
            got = self.allgather(obj)
            return got if self.rank == dst else None
        if self.rank == 0:
            out = [obj] + [_recv(c) for c in self.peers]
            for c in self.peers:
                _send(c, None)
            return out
        _send(self.sock, obj)
        _recv(self.sock)
        return None

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather(None)

    def max(self, x):
        return max(self.allgather(float(x)))

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None


class GlooGroup:
    """the same interface over torch.distributed (any initialised backend; the CPU tests use gloo)"""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def gather(self, obj, dst=0):
        out = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(obj, out, dst=dst)
        return out

    def broadcast(self, obj, src=0):
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self):
        self.dist.barrier()

    def max(self, x):
        return max(self.allgather(float(x)))

    def close(self):
        pass
