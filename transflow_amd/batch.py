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
import json
import os
import secrets
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


def pass_pairs(total_pairs: int, batch: int, world: int, equal: bool = False) -> list[int]:
    """Pairs per pass of every rank: min(batch, the rank's shard).  Shards differ by at most one pair, so at
    world = 8 over the 255 pairs of a T = 256 clip seven ranks run 32 pairs per pass and the last one 31.
    `equal` trims every rank to the smallest of them (each pass then moves the same bytes on every rank)."""
    sizes = []
    for r in range(world):
        a, b = shard_range(total_pairs, r, world)
        sizes.append(min(int(batch), b - a))
    if equal and sizes:
        sizes = [min(sizes)] * world
    return sizes


def gather_counts(pairs_per_pass: list[int], frame_bytes: int) -> tuple[list[int], list[int]]:
    """(recv_bytes, offsets) of the ONE tf_batch_gather a pass ends with: rank r sends the
    pairs_per_pass[r] frames of its pass (side by side in one buffer) and they land on the root at
    offsets[r].  Every rank derives the same lists from the plans, so every Recv the root posts has its
    Send even when the ranks' passes differ in length."""
    counts = [int(n) * int(frame_bytes) for n in pairs_per_pass]
    offsets, off = [], 0
    for c in counts:
        offsets.append(off)
        off += c
    return counts, offsets


def gather_calls(pairs_per_pass: list[int], frame_bytes: int, root: int = 0) -> list[dict]:
    """The argument list each rank hands to tf_batch_gather for one pass (what bench.py's gather leg does,
    and what tests/test_batch_gloo.py replays against csrc/batch.hip's posting rules)."""
    counts, _ = gather_counts(pairs_per_pass, frame_bytes)
    total = sum(counts)
    return [{"rank": r, "send_bytes": counts[r], "recv_bytes": counts if r == root else None,
             "recv_capacity": total if r == root else 0, "root": root} for r in range(len(counts))]


def flows_to_root_calls(plans: list[dict], flow_bytes: int, root: int = 0) -> list[list[dict]]:
    """"Flows to root" (mode F): the argument lists of the tf_batch_gather_at calls that bring the
    whole clip's flows to one rank in clip order, so that ONE compositor consumes them as the reference's does
    (transflow/pipeline.py:565; the remap is a recurrence over the clip, compositor/layers/movement.py:51-52, so
    per-rank streams are not the reference's frames, the root's are).  One call per pass index k on every
    rank: rank r sends the flows of its pass k (pairs_per_pass[r] of them; nothing when it has no pass k) and they
    land at the clip position of that pass's first pair, plans[r]["pairs"][0] + plans[r]["pass_starts"][k].  A rank's
    last pass may repeat a few pairs of the one before (batch_starts): the same flows land on the same place.
    Every rank derives the same lists from the all-gathered plans.  -> calls[k][r]."""
    total = max((p["pairs"][1] for p in plans), default=0) - min((p["pairs"][0] for p in plans), default=0)
    first = min((p["pairs"][0] for p in plans), default=0)
    n_pass = max((len(p["pass_starts"]) for p in plans), default=0)
    calls = []
    for k in range(n_pass):
        counts, offsets = [], []
        for p in plans:
            has = k < len(p["pass_starts"])
            counts.append(int(p["pairs_per_pass"]) * int(flow_bytes) if has else 0)
            offsets.append((p["pairs"][0] - first + p["pass_starts"][k]) * int(flow_bytes) if has else 0)
        calls.append([{"rank": r, "send_bytes": counts[r], "recv_bytes": counts if r == root else None,
                       "recv_offsets": offsets if r == root else None,
                       "recv_capacity": total * int(flow_bytes) if r == root else 0, "root": root}
                      for r in range(len(plans))])
    return calls


# ---- rendezvous ------------------------------------------------------------------------------------
def _private_dir() -> str:
    """A directory only this user can enter (0700, owned by us, not a symlink): the rendezvous file of a launch
    that no launcher named lives here, so no other local user can plant or replace it."""
    d = os.path.join(tempfile.gettempdir(), f"tfhip-{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f"rendezvous directory {d} is not a private directory of this user")
    return d


def rendezvous_path() -> str:
    """One file per launch.  bench.py's launcher names it (TF_BATCH_RDZV, inside a fresh 0700 mkdtemp
    directory); under torch.distributed.run all ranks are children of one agent process, whose pid with
    the master port and restart count identifies the launch, inside this user's private directory."""
    p = os.environ.get("TF_BATCH_RDZV")
    if p:
        return p
    key = "-".join([os.environ.get("MASTER_PORT", "0"), str(os.getppid()),
                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                    "".join(c for c in os.environ.get("TORCHELASTIC_RUN_ID", "none") if c.isalnum())[:32]])
    return os.path.join(_private_dir(), f"rdzv-{key}")


def _encode(obj):
    """Host messages are JSON (ints, floats, strings, None, lists, dicts); bytes travel as hex under a tag.
    Nothing that arrives on the socket is ever unpickled."""
    if isinstance(obj, (bytes, bytearray)):
        return {"__bytes__": bytes(obj).hex()}
    if isinstance(obj, dict):
        for k in obj:
            if not isinstance(k, str):        # JSON would turn the key into a string and the receiver would see another dict
                raise TypeError(f"HostGroup carries dicts with string keys only, not {type(k).__name__} keys")
        return {k: _encode(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):        # (a tuple arrives as a list)
        return [_encode(v) for v in obj]
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if getattr(obj, "ndim", None) == 0 and hasattr(obj, "item"):   # numpy scalars (an array of any size is not one)
        return _encode(obj.item())
    raise TypeError(f"HostGroup cannot carry a {type(obj).__name__}")


def _decode(obj):
    if isinstance(obj, dict):
        if set(obj) == {"__bytes__"}:
            return bytes.fromhex(obj["__bytes__"])
        return {k: _decode(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_decode(v) for v in obj]
    return obj


MAX_MESSAGE = 1 << 26


def _send_msg(sock: socket.socket, obj) -> None:
    data = json.dumps(_encode(obj), allow_nan=True).encode()
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
    if n > MAX_MESSAGE:
        raise ConnectionError(f"rendezvous message of {n} bytes refused")
    return _decode(json.loads(_recv_exact(sock, n)))


class HostGroup:
    """Host-side star of the ranks of one node: rank 0 listens on an ephemeral port of 127.0.0.1
    and publishes it, with two random tokens, through the rendezvous file (created exclusively, mode
    0600, never through a symlink); every collective is gather-to-0 + answer.  A peer must present the
    first token before anything it sends is parsed, and rank 0 answers with the second, so neither side
    talks to a process that could not read the file."""

    def __init__(self, rank: int | None = None, world: int | None = None, path: str | None = None,
                 timeout: float = 300.0):
        r, _, w = env_world()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.timeout = timeout
        self.peers: list[socket.socket | None] = []
        self.sock: socket.socket | None = None
        if self.world == 1:
            self.path = path
            return
        self.path = path or rendezvous_path()
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            srv.settimeout(timeout)
            hello, answer = secrets.token_hex(16), secrets.token_hex(16)
            tmp = f"{self.path}.{os.getpid()}.tmp"
            try:
                os.unlink(tmp)                  # a leftover of a process that died with this pid
            except OSError:
                pass
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
            with os.fdopen(fd, "w") as f:
                f.write(f"127.0.0.1:{srv.getsockname()[1]} {hello} {answer}\n")
            os.replace(tmp, self.path)          # atomic: a reader sees nothing or the whole line
            self.peers = [None] * self.world
            deadline = time.monotonic() + timeout
            joined = 0
            while joined < self.world - 1:
                if time.monotonic() > deadline:
                    raise TimeoutError("rendezvous: not every rank arrived")
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(1.0)            # a local peer sends its 36 bytes at once; a silent one holds the loop a second
                try:
                    head = _recv_exact(conn, len(hello) + 4)
                    token, peer = head[:len(hello)].decode("ascii", "replace"), struct.unpack("<i", head[len(hello):])[0]
                    if not secrets.compare_digest(token, hello) or not (0 < peer < self.world):
                        raise ConnectionError("bad token or rank")
                    conn.sendall(answer.encode())
                except (OSError, ConnectionError, struct.error):
                    conn.close()                # a stranger on the port: dropped, nothing of it was parsed
                    continue
                conn.settimeout(timeout)
                if self.peers[peer] is not None:
                    # the rank knows the token and comes again: its first attempt failed on its side (a timeout, a reset)
                    # after we had accepted it -- the new connection takes the old one's place
                    self.peers[peer].close()
                else:
                    joined += 1
                self.peers[peer] = conn
            srv.close()
            try:
                os.unlink(self.path)            # everyone is in: the file has done its job
            except OSError:
                pass
        else:
            # A file left behind by a launch that died (same port, same parent) may still name a dead listener, and a
            # listener that is not this launch's does not know its token: keep reading and trying until the deadline.
            deadline = time.monotonic() + timeout
            last = "no rendezvous file"
            while self.sock is None:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rendezvous through {self.path} failed ({last})")
                try:
                    fd = os.open(self.path, os.O_RDONLY | os.O_NOFOLLOW)
                    with os.fdopen(fd) as f:
                        st = os.fstat(f.fileno())
                        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
                            raise RuntimeError(f"rendezvous file {self.path} is not a private file of this user")
                        line = f.read().strip()
                    where, hello, answer = line.split(" ")
                    host, port = where.rsplit(":", 1)
                    if host != "127.0.0.1":
                        raise RuntimeError(f"rendezvous file {self.path} names {host}, not this node")
                    sock = socket.create_connection((host, int(port)), timeout=min(timeout, 30.0))
                    try:
                        sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        sock.sendall(hello.encode() + struct.pack("<i", self.rank))
                        if not secrets.compare_digest(_recv_exact(sock, len(answer)).decode("ascii", "replace"), answer):
                            raise ConnectionError("the listener did not know the launch's token")
                    except BaseException:
                        sock.close()
                        raise
                    sock.settimeout(timeout)
                    self.sock = sock
                except (OSError, ValueError, ConnectionError) as err:
                    last = f"{type(err).__name__}: {err}"
                    time.sleep(0.05)

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

    def gather_at(self, send_ptr: int, send_bytes: int, recv_ptr: int | None = None, recv_bytes=None, recv_offsets=None,
                  recv_capacity: int = 0, root: int = 0) -> None:
        """tf_batch_gather_at: rank r's bytes land on root at recv_ptr + recv_offsets[r] (flows_to_root_calls)."""
        counts = offs = None
        if recv_bytes is not None:
            counts = (C.c_size_t * self.world)(*[int(v) for v in recv_bytes])
            offs = (C.c_size_t * self.world)(*[int(v) for v in recv_offsets])
        self._check(self._lib.tf_batch_gather_at(self._h, C.c_void_p(send_ptr) if send_ptr else None, int(send_bytes),
                                                 C.c_void_p(recv_ptr) if recv_ptr else None, counts, offs,
                                                 int(recv_capacity), int(root)))

    def gather_begin(self, send_ptr: int, send_bytes: int, recv_ptr: int | None = None, recv_bytes=None, root: int = 0) -> None:
        """gather_dev beside what the library stream does next (tf_batch_gather_begin); gather_end() before the send
        buffer is written again."""
        counts = None
        if recv_bytes is not None:
            counts = (C.c_size_t * self.world)(*[int(v) for v in recv_bytes])
        self._check(self._lib.tf_batch_gather_begin(self._h, C.c_void_p(send_ptr) if send_ptr else None, int(send_bytes),
                                                    C.c_void_p(recv_ptr) if recv_ptr else None, counts, int(root)))

    def gather_end(self) -> None:
        self._check(self._lib.tf_batch_gather_end(self._h))

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
