"""Batch-of-frames mode: shard independent frame pairs over the GPUs of one node.

With flags == 0 every Farnebäck pair is an independent computation (reference
transflow/flow/sources/cv.py:478-490: the `flow=` argument is only an output
buffer), so rank r of R owns a contiguous range of pairs plus a one-frame halo
(SURVEY.md §8e).  The remap recurrence is serial per stream
(compositor/layers/movement.py:51-52), so each rank runs it over its own range
("independent streams"); the path has no data-path collective.

Two transports, neither of them torch:

* `HostGroup` -- the ranks of one node meet through a rendezvous file and a TCP
  star on 127.0.0.1 (rank 0 listens).  It carries the 128-byte RCCL id, the
  barrier around the timed region, max/sum of a few host numbers and small host
  gathers (per-rank results for the JSON line).  Pure Python; the CPU tests run it.
* `RcclGroup` -- tf_batch_* of libtfhip.so: RCCL over xGMI on device buffers, for
  the one-off broadcast of the shared inputs (pixmap, reset mask) and the gather
  of finished frames to rank 0.
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
import socket
import struct
import tempfile
import time


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced split of `total` units: ranks < total % world get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(total), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def frames_needed(pair_range: tuple[int, int]) -> tuple[int, int]:
    """Frame indices a rank must hold for its pairs: pair t uses frames t and t+1."""
    a, b = pair_range
    return (a, b + 1) if b > a else (a, a)


def batch_starts(n_pairs: int, batch: int) -> list[int]:
    """Where the passes over a shard of `n_pairs` pairs start, `batch` pairs each: consecutive
    windows, the last one moved back so that it ends with the shard (it then repeats a few pairs
    rather than running a short batch).  A shard shorter than `batch` is one short pass."""
    if n_pairs <= 0:
        return []
    if n_pairs <= batch:
        return [0]
    starts = list(range(0, n_pairs - batch + 1, batch))
    if starts[-1] + batch < n_pairs:
        starts.append(n_pairs - batch)
    return starts


def env_world():
    """(rank, local_rank, world_size) as torch.distributed.run (and bench.py's own launcher) export them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def rendezvous_path() -> str:
    """One file per launch.  bench.py's launcher names it (TF_BATCH_RDZV); under
    torch.distributed.run all ranks are children of one agent process, whose pid with the
    master port and restart count identifies the launch."""
    p = os.environ.get("TF_BATCH_RDZV")
    if p:
        return p
    key = "-".join([os.environ.get("MASTER_PORT", "0"), str(os.getppid()),
                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                    os.environ.get("TORCHELASTIC_RUN_ID", "none")])
    return os.path.join(tempfile.gettempdir(), f"tfhip-rdzv-{key}")


def _send_msg(sock: socket.socket, obj) -> None:
    data = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock: socket.socket):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


class HostGroup:
    """Host-side star of the ranks of one node: rank 0 listens on an ephemeral port of 127.0.0.1
    and publishes it through the rendezvous file; every collective is gather-to-0 + answer."""

    def __init__(self, rank: int | None = None, world: int | None = None, path: str | None = None,
                 timeout: float = 300.0):
        r, _, w = env_world()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.path = path or rendezvous_path()
        self.timeout = timeout
        self.peers: list[socket.socket | None] = []
        self.sock: socket.socket | None = None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            srv.settimeout(timeout)
            tmp = f"{self.path}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                f.write(f"127.0.0.1:{srv.getsockname()[1]}\n")
            os.replace(tmp, self.path)          # atomic: a reader sees nothing or the whole line
            self.peers = [None] * self.world
            for _ in range(self.world - 1):
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                peer = _recv_msg(conn)
                if not (0 < peer < self.world) or self.peers[peer] is not None:
                    raise RuntimeError(f"rendezvous: unexpected rank {peer}")
                self.peers[peer] = conn
            srv.close()
            try:
                os.unlink(self.path)            # everyone is in: the file has done its job
            except OSError:
                pass
        else:
            deadline = time.monotonic() + timeout
            addr = None
            while addr is None:
                try:
                    with open(self.path) as f:
                        line = f.read().strip()
                    host, port = line.rsplit(":", 1)
                    addr = (host, int(port))
                except (OSError, ValueError):
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rank {self.rank}: no rendezvous file {self.path}")
                    time.sleep(0.02)
            self.sock = socket.create_connection(addr, timeout=timeout)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send_msg(self.sock, self.rank)

    # -- collectives on host objects ---------------------------------------------------
    def gather(self, obj):
        """List of every rank's `obj` on rank 0, None elsewhere."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [None] * (self.world - 1)
            for r in range(1, self.world):
                out[r] = _recv_msg(self.peers[r])
            return out
        _send_msg(self.sock, obj)
        return None

    def broadcast(self, obj=None):
        """Rank 0's `obj` on every rank."""
        if self.world == 1:
            return obj
        if self.rank == 0:
            for r in range(1, self.world):
                _send_msg(self.peers[r], obj)
            return obj
        return _recv_msg(self.sock)

    def allgather(self, obj):
        return self.broadcast(self.gather(obj))

    def barrier(self) -> None:
        self.allgather(None)

    def max_over_ranks(self, value: float) -> float:
        return max(self.allgather(float(value)))

    def sum_over_ranks(self, value: float) -> float:
        return sum(self.allgather(float(value)))

    def close(self) -> None:
        for s in self.peers + [self.sock]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self.peers, self.sock = [], None


class RcclGroup:
    """tf_batch_* of libtfhip.so: the RCCL communicator of this launch's ranks, on the device
    tf_init selected.  The id travels through `host` (rank 0 makes it)."""

    def __init__(self, host: HostGroup):
        from . import _lib
        self._lib = _lib.load()
        self._check = _lib.check
        self.rank, self.world = host.rank, host.world
        uid = (C.c_uint8 * 128)()
        if self.rank == 0:
            self._check(self._lib.tf_batch_unique_id(uid))
        raw = host.broadcast(bytes(uid) if self.rank == 0 else None)
        uid = (C.c_uint8 * 128).from_buffer_copy(raw)
        self._h = C.c_void_p()
        self._check(self._lib.tf_batch_init(C.byref(self._h), self.rank, self.world, uid))
        v = C.c_int()
        self._check(self._lib.tf_batch_info(self._h, None, None, C.byref(v)))
        self.rccl_version = v.value

    def broadcast_dev(self, dev_ptr: int, nbytes: int, root: int = 0) -> None:
        self._check(self._lib.tf_batch_broadcast(self._h, C.c_void_p(dev_ptr), int(nbytes), int(root)))

    def gather_dev(self, send_ptr: int, send_bytes: int, recv_ptr: int | None = None, recv_bytes=None,
                   root: int = 0) -> None:
        counts = None
        if recv_bytes is not None:
            counts = (C.c_size_t * self.world)(*[int(v) for v in recv_bytes])
        self._check(self._lib.tf_batch_gather(self._h, C.c_void_p(send_ptr) if send_ptr else None, int(send_bytes),
                                              C.c_void_p(recv_ptr) if recv_ptr else None, counts, int(root)))

    def reduce(self, values, op: str = "max"):
        vals = [float(v) for v in values]
        arr = (C.c_double * max(1, len(vals)))(*vals)
        self._check(self._lib.tf_batch_reduce(self._h, arr, len(vals), {"sum": 0, "max": 1}[op]))
        return [arr[i] for i in range(len(vals))]

    def barrier(self) -> None:
        """Every rank's library stream has drained and every rank has arrived."""
        self.reduce([], "max")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tf_batch_destroy(self._h)
            self._h = C.c_void_p()

    def abandon(self) -> None:
        """Forget the communicator without destroying it (a peer never joined: ncclCommDestroy could wait for it)."""
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def launch_ranks(argv: list[str], world: int, env: dict | None = None, timeout: float | None = None) -> int:
    """Starts `world` fresh rank processes of `argv` (a full command line) on this node, rank r on
    GPU r, and waits for them.  Must be called from a process that has made no GPU call: the ranks
    are children started with subprocess, never an exec of a process that touched the GPU.  Returns
    the largest exit code; the ranks inherit stdout/stderr (rank 0 prints the result line)."""
    import subprocess
    rdzv = os.path.join(tempfile.mkdtemp(prefix="tfhip-launch-"), "rdzv")
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                 TF_BATCH_RDZV=rdzv)
        e.setdefault("MASTER_ADDR", "127.0.0.1")
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what this pool's driver supports
        procs.append(subprocess.Popen(argv, env=e))
    rc = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                rc = max(rc, abs(code))
                if code != 0:                       # one rank failed: the others would wait forever
                    for q in pending:
                        q.terminate()
            if deadline is not None and time.monotonic() > deadline:
                for q in pending:
                    q.terminate()
                rc = max(rc, 124)
                deadline = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink(rdzv)
        except OSError:
            pass
        try:
            os.rmdir(os.path.dirname(rdzv))
        except OSError:
            pass
    return rc
