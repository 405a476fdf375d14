"""A ~100-line control plane for one-process-per-GPU runs: no torch, no MPI.

The reference leaves process management to MPI inside its third-party samplers
(evidence/polychord/__init__.py:21-29,176-199; evidence/ultranest/__init__.py:21-29,151-194).  Here the data
path between GPUs is RCCL (rvll_allgather_*); what the ranks need besides is tiny and infrequent — hand the
128-byte communicator id from rank 0 to everybody, a barrier around the timed region, a max / min over ranks, and
(for samplers that shard host arrays) an all-gather of small buffers.  That is a star over stream sockets with
rank 0 in the middle:

    Rendezvous.from_env()         RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as the launcher exports them
    .broadcast(obj, src=0)        picklable object from rank src to every rank
    .allgather(obj)               list of every rank's object, in rank order, on every rank
    .barrier()
    .allreduce(x, op)             op in {"max", "min", "sum"} over python numbers

Address: RVLL_RDZV=tcp://host:port or unix:name if set.  Otherwise an abstract unix socket named after
MASTER_ADDR, MASTER_PORT and the launcher's run id — `python -m torch.distributed.run` keeps its own store
LISTENING on MASTER_PORT, so on the one node it launches for, the ranks meet beside it, not on it.
Importing this module must not import torch: with torch loaded first a process binds torch's bundled HIP
runtime and RCCL instead of the ROCm ones librvll.so is built against.
"""
import os
import pickle
import socket
import struct
import time


class RendezvousError(RuntimeError):
    pass


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("!Q", len(data)) + data)


def _recv(sock):
    head = _recv_exact(sock, 8)
    return pickle.loads(_recv_exact(sock, struct.unpack("!Q", head)[0]))


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        buf += chunk
    return bytes(buf)


def default_address(env=os.environ):
    if env.get("RVLL_RDZV"):
        return env["RVLL_RDZV"]
    run = env.get("TORCHELASTIC_RUN_ID", "none")
    return f"unix:rvll-rdzv-{env.get('MASTER_ADDR', '127.0.0.1')}-{env.get('MASTER_PORT', '29500')}-{run}"


def _open(address, listen):
    kind, _, rest = address.partition(":")
    if kind == "unix":
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        target = "\0" + rest                               # abstract namespace: nothing to unlink, gone with the process
    elif kind == "tcp":
        host, _, port = rest.lstrip("/").rpartition(":")
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        target = (host, int(port))
    else:
        raise RendezvousError(f"unknown rendezvous address {address!r}")
    if listen:
        s.bind(target)
        s.listen(1024)
    return s, target


class Rendezvous:
    def __init__(self, rank, world, address=None, timeout=120.0):
        if not 0 <= rank < world:
            raise ValueError("0 <= rank < world required")
        self.rank, self.world, self.timeout = rank, world, timeout
        self.address = address or default_address()
        self._peers = {}                                   # rank 0: rank -> socket
        self._hub = None                                   # other ranks: socket to rank 0
        if world == 1:
            return
        if rank == 0:
            srv, _ = _open(self.address, listen=True)
            srv.settimeout(timeout)
            try:
                while len(self._peers) < world - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(timeout)
                    peer = _recv(conn)
                    if not isinstance(peer, int) or not 0 < peer < world or peer in self._peers:
                        conn.close()
                        raise RendezvousError(f"unexpected peer announcement {peer!r}")
                    self._peers[peer] = conn
            except socket.timeout as exc:
                raise RendezvousError(f"only {len(self._peers) + 1} of {world} ranks arrived at {self.address}") from exc
            finally:
                srv.close()
        else:
            deadline = time.monotonic() + timeout
            while True:
                s, target = _open(self.address, listen=False)
                try:
                    s.connect(target)
                    break
                except (ConnectionRefusedError, FileNotFoundError, OSError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise RendezvousError(f"rank 0 is not listening at {self.address}")
                    time.sleep(0.05)
            s.settimeout(timeout)
            _send(s, rank)
            self._hub = s

    @classmethod
    def from_env(cls, env=os.environ, **kw):
        return cls(int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), **kw)

    def allgather(self, obj):
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            items = [obj] + [None] * (self.world - 1)
            for r, s in self._peers.items():
                items[r] = _recv(s)
            for s in self._peers.values():
                _send(s, items)
            return items
        _send(self._hub, obj)
        return _recv(self._hub)

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather(None)

    def allreduce(self, x, op="max"):
        vals = self.allgather(x)
        return {"max": max, "min": min, "sum": sum}[op](vals)

    def close(self):
        for s in list(self._peers.values()) + ([self._hub] if self._hub else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._hub = {}, None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
