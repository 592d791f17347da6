"""Minimal TCP rendezvous for the multi-GPU bench (stdlib sockets, star topology through rank 0).

Why not torch.distributed: PyTorch's ROCm wheel bundles its own libamdhip64.so / libhsa-runtime64.so (ROCm 7.0, no
SONAME) while libse2gpu and the system RCCL use /opt/rocm (7.2).  Merely importing torch puts a second, un-initialised
HSA runtime into the process under the name RCCL dlopen()s, and RCCL then fails with "no ROCm-capable device".  The
data path needs one thing from the outside world - the 128-byte ncclUniqueId on every rank - plus a barrier and a
max() for the timings; that is what this file provides.  Reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT like
torch.distributed would (the port used is MASTER_PORT + 1: torchrun's own store listens on MASTER_PORT).
"""
from __future__ import annotations

import os
import socket
import struct
import time


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return buf


class Rendezvous:
    def __init__(self, rank: int, world: int, addr: str | None = None, port: int | None = None, timeout: float = 120.0):
        self.rank, self.world = rank, world
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + 1
        self.peers: list[socket.socket] = []
        self.sock: socket.socket | None = None
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
            srv.listen(world)
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = struct.unpack("<i", _recv_exact(c, 4))[0]
                conns[r] = c
            srv.close()
            self.peers = [conns[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack("<i", rank))
            self.sock = s

    def broadcast(self, data: bytes | None, nbytes: int) -> bytes:
        """rank 0's `data` (nbytes long) to every rank"""
        if self.world == 1:
            return bytes(data)
        if self.rank == 0:
            for p in self.peers:
                p.sendall(data)
            return bytes(data)
        return _recv_exact(self.sock, nbytes)

    def allreduce_max(self, value: float) -> float:
        if self.world == 1:
            return value
        if self.rank == 0:
            vals = [value] + [struct.unpack("<d", _recv_exact(p, 8))[0] for p in self.peers]
            m = max(vals)
            for p in self.peers:
                p.sendall(struct.pack("<d", m))
            return m
        self.sock.sendall(struct.pack("<d", value))
        return struct.unpack("<d", _recv_exact(self.sock, 8))[0]

    def allreduce_sum(self, value: float) -> float:
        """the sum over ranks, added in rank order on rank 0 (every rank gets the same bits)"""
        if self.world == 1:
            return value
        if self.rank == 0:
            total = value
            for p in self.peers:
                total += struct.unpack("<d", _recv_exact(p, 8))[0]
            for p in self.peers:
                p.sendall(struct.pack("<d", total))
            return total
        self.sock.sendall(struct.pack("<d", value))
        return struct.unpack("<d", _recv_exact(self.sock, 8))[0]

    def barrier(self):
        self.allreduce_max(0.0)

    def close(self):
        for p in self.peers:
            p.close()
        if self.sock:
            self.sock.close()
