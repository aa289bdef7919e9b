"""Minimal TCP rendezvous for one-process-per-GPU runs (bench.py, multi-GPU drivers).

Only three things are needed outside the data path: hand the RCCL unique id from rank 0 to
the others, a barrier, and a max-reduction of the measured time.  They are done with plain
sockets (rank 0 listens on an ephemeral port published through a file in /tmp named after
MASTER_PORT and the launcher's run id; the launcher's own store owns MASTER_PORT).

Why not torch.distributed: importing torch pulls the wheel's bundled ROCm runtime
(libamdhip64 / libhsa-runtime64 / librccl of another ROCm release) into the process next to
the system libraries libjtprop.so links, and RCCL's communicator init then fails with
"no ROCm-capable device is detected" (observed on MI355X, ROCm 7.2 + torch 2.10+rocm7.0).
The separator messages themselves never touch this channel: they move GPU to GPU with
ncclSend / ncclRecv inside libjtprop.so.
"""

import os
import socket
import struct
import time

__all__ = ["Rendezvous"]


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<I", len(payload)) + payload)


def _recv_msg(sock):
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    return _recv_exact(sock, n)


class Rendezvous:
    """Star topology: rank 0 accepts one connection per other rank."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29501, timeout=120.0, port_file=None):
        """`port_file` (single node): rank 0 binds an ephemeral port and publishes it in that file,
        so a busy MASTER_PORT + 1 cannot break the run; without it the fixed `port` is used."""
        self.rank, self.world = rank, world
        self.peers = []
        self.sock = None
        self._port_file = port_file
        if world <= 1:
            return
        host = "127.0.0.1" if addr in ("localhost", "") else addr
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            if port_file is not None:
                srv.bind((host, 0))
                tmp = "%s.%d" % (port_file, os.getpid())
                with open(tmp, "w") as fh:
                    fh.write("%d\n" % srv.getsockname()[1])
                os.replace(tmp, port_file)
            else:
                srv.bind((host, port))
            srv.listen(world)
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                (peer,) = struct.unpack("<I", _recv_exact(conn, 4))
                conns[peer] = conn
            srv.close()
            self.peers = [conns[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    if port_file is not None:
                        with open(port_file) as fh:
                            port = int(fh.read().strip())
                    s = socket.create_connection((host, port), timeout=5.0)
                    break
                except (OSError, ValueError):       # not published yet, or a stale file of an earlier run
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack("<I", rank))
            self.sock = s

    def _port_file_cleanup(self):
        path = getattr(self, "_port_file", None)
        if path and self.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass

    def broadcast(self, payload=None):
        """bytes from rank 0 to everyone; returns the payload on every rank."""
        if self.world <= 1:
            return payload
        if self.rank == 0:
            for p in self.peers:
                _send_msg(p, payload)
            return payload
        return _recv_msg(self.sock)

    def allreduce_max(self, value):
        if self.world <= 1:
            return value
        if self.rank == 0:
            vals = [value] + [struct.unpack("<d", _recv_msg(p))[0] for p in self.peers]
            out = max(vals)
            for p in self.peers:
                _send_msg(p, struct.pack("<d", out))
            return out
        _send_msg(self.sock, struct.pack("<d", value))
        return struct.unpack("<d", _recv_msg(self.sock))[0]

    def allgather(self, payload):
        """bytes from every rank -> the list of all ranks' payloads (rank order), on every rank."""
        if self.world <= 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(p) for p in self.peers]
            blob = b"".join(struct.pack("<I", len(x)) + x for x in parts)
            for p in self.peers:
                _send_msg(p, blob)
        else:
            _send_msg(self.sock, payload)
            blob = _recv_msg(self.sock)
        out, at = [], 0
        while at < len(blob):
            (n,) = struct.unpack("<I", blob[at:at + 4])
            out.append(blob[at + 4:at + 4 + n])
            at += 4 + n
        return out

    def barrier(self):
        self.allreduce_max(0.0)

    def close(self):
        self._port_file_cleanup()
        for p in self.peers:
            p.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None
