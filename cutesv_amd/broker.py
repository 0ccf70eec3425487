"""One HIP context per GPU for a whole pool of cuteSV workers.

The reference's phase 3 (main script :1113-1199) forks `Pool(processes=threads)` and hands every worker one (chromosome, type)
task at a time.  A HIP context cannot be shared across processes and costs each process its own runtime start, device arenas
and page-locked landing zones; SURVEY.md 8(b) "Caller" therefore puts ONE process per GPU under the five task callables.
This module is that process and the workers' side of it:

  * `python -m cutesv_amd.broker --name N --device D --watch-pid P` owns the only `engine.Context` of the GPU.  It listens on
    an abstract unix socket (nothing on disk, gone with the process), accepts the pool's workers, and runs their
    `csv_cluster_batch` calls.  Requests that are waiting at the same moment become ONE batch - their segments side by side
    in the context's page-locked staging columns, one H2D, one launch sequence, one D2H - and every requester gets its own
    slice of the structure of arrays back.
  * `Client` is what a worker holds instead of a context (`resolve.context()`): the task's columns and its caller-owned
    result arrays live in one anonymous shared-memory region (`memfd_create`, passed once over the socket), a request is the
    task's `csv_batch_in` / `csv_batch_out` structs with their pointers expressed in that region.  Pickle walking before
    and row building after the call stay in the worker: that is the part of a task that scales with the pool.

Who starts it: `resolve.warm_up()` in the pool's parent (before `Pool(...)`: the runtime start then overlaps cuteSV's
extraction phase), otherwise the first worker that finds nobody listening (a lock file keeps it to one).  It exits when the
watched process is gone, on `shutdown()`, or after `--linger` seconds without a client.

No CPU path: the broker fails to start without libcutesv_hip.so and a GPU, and a worker whose broker is gone raises.
"""
import ctypes as C
import mmap
import os
import selectors
import socket
import struct
import sys
import time

import numpy as np

from . import _abi

MAGIC = 0x42565343                                   # "CSVB"
HDR = struct.Struct("<IIQ")                          # magic, kind, payload bytes
K_MAP, K_CALL, K_INFO, K_SHUTDOWN, K_REPLY, K_STATS, K_PUT, K_GET, K_FLUSH = 1, 2, 3, 4, 5, 6, 7, 8, 9
REPLY = struct.Struct("<iqqqqq")                     # rc, cap_calls, cap_support, n_calls, n_support, n_clusters  (+ error text)
ALIGN = 256


class BrokerError(RuntimeError):
    pass


def socket_name(owner_pid, device):
    """abstract socket of the broker of (pool parent, device); CUTESV_AMD_BROKER_NAME (set by resolve.warm_up) overrides the prefix"""
    prefix = os.environ.get("CUTESV_AMD_BROKER_NAME") or "cutesv_amd-%d-%d" % (os.getuid(), owner_pid)
    return "%s-gpu%d" % (prefix, device)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise EOFError("peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    """-> (kind, payload, fds); descriptors arrive with the first byte of the header"""
    data, fds, _flags, _addr = socket.recv_fds(sock, HDR.size, 4)
    if not data:
        raise EOFError("peer closed the connection")
    if len(data) < HDR.size:
        data += _recv_exact(sock, HDR.size - len(data))
    magic, kind, n = HDR.unpack(data)
    if magic != MAGIC:
        raise BrokerError("bad message header")
    return kind, (_recv_exact(sock, n) if n else b""), fds


def _send_msg(sock, kind, payload=b"", fds=()):
    msg = HDR.pack(MAGIC, kind, len(payload)) + payload
    if fds:
        sent = socket.send_fds(sock, [msg], list(fds))
        if sent < len(msg):
            sock.sendall(msg[sent:])
    else:
        sock.sendall(msg)


def _pointer_fields(struct_type):
    return [name for name, tp in struct_type._fields_ if tp is C.c_void_p]


_IN_PTRS = _pointer_fields(_abi.BatchIn)
_OUT_PTRS = _pointer_fields(_abi.BatchOut)


# =============================================================================================== worker side
class Region:
    """Anonymous shared memory (memfd) with a bump allocator: a task's columns and result arrays.  Grow-only; a new
    descriptor is made (and sent to the broker) when a task needs more."""

    def __init__(self, nbytes):
        self.size = max(int(nbytes), 1 << 20)
        self.fd = os.memfd_create("cutesv_amd", 0)
        os.ftruncate(self.fd, self.size)
        self.mm = mmap.mmap(self.fd, self.size)
        self.base = C.addressof(C.c_char.from_buffer(self.mm))
        self.top = 0

    def reset(self):
        self.top = 0

    def alloc(self, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) if not isinstance(shape, (int, np.integer)) else int(shape)
        off = (self.top + ALIGN - 1) // ALIGN * ALIGN
        end = off + n * dtype.itemsize
        if end > self.size:
            raise MemoryError("shared region too small")          # (Client sizes the region before it allocates)
        self.top = end
        return np.frombuffer(self.mm, dtype=dtype, count=n, offset=off).reshape(shape)

    def put(self, arr, dtype=None):
        out = self.alloc(arr.shape, dtype or arr.dtype)
        np.copyto(out, arr, casting="unsafe")         # (a narrowing copy only after _fits32 said the values fit)
        return out

    def close(self):
        try:
            os.close(self.fd)
        except OSError:
            pass
        self.fd = -1                                             # (the mapping goes with its last numpy view)


def _fits32(x, y):
    """both columns exist and hold int32 values (already narrow, or int64 within range)"""
    if x is None or y is None:
        return False
    for v in (x, y):
        if v.dtype == np.int32:
            continue
        if v.dtype != np.int64:
            return False
        if len(v) and (int(v.min()) < -(1 << 31) or int(v.max()) >= (1 << 31)):
            return False
    return True


def _padded(nbytes):
    return (int(nbytes) + ALIGN - 1) // ALIGN * ALIGN + ALIGN


class Client:
    """What a pool worker holds in place of an `engine.Context`: `cluster_batch(batch, ...)` with the same arguments and the
    same `HostResult` back, computed by the GPU's broker."""

    def __init__(self, sock, device, name):
        self.sock, self.device, self.name = sock, device, name
        self.region = None
        self.calls = 0

    # ---- connecting / starting
    @classmethod
    def connect(cls, device=0, owner_pid=None, spawn=True, timeout=120.0):
        """the broker of (`owner_pid`'s pool, device); spawn=True starts one if nobody listens (one starter per broker: a lock file)"""
        if owner_pid is None:
            import multiprocessing as mp
            pp = mp.parent_process()
            owner_pid = pp.pid if pp is not None else os.getpid()
        name = socket_name(owner_pid, device)
        s = _try_connect(name)
        if s is None:
            if not spawn:
                raise BrokerError("no broker is listening on %r" % name)
            s = _spawn_and_connect(name, device, owner_pid, timeout)
        s.settimeout(float(os.environ.get("CUTESV_AMD_BROKER_TIMEOUT", "900")))
        return cls(s, device, name)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass
        if self.region is not None:
            self.region.close()
            self.region = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _request(self, kind, payload=b"", fds=()):
        try:
            _send_msg(self.sock, kind, payload, fds)
            k, data, _ = _recv_msg(self.sock)
        except (OSError, EOFError) as e:
            raise BrokerError("the GPU broker %r went away (%s); there is no CPU path" % (self.name, e)) from e
        if k != K_REPLY:
            raise BrokerError("unexpected message %d from the broker" % k)
        return data

    def info(self):
        """dict(pid, device, bus, compute_units, calls, batches, ...) of the broker"""
        import json
        return json.loads(self._request(K_INFO).decode())

    def shutdown(self):
        self._request(K_SHUTDOWN)

    # ---- a chromosome's walked reads block, shared between the pool's workers (columns.WalkedReads)
    def reads_get(self, key):
        """the block another worker walked and left with the broker, or None"""
        from .columns import WalkedReads
        try:
            _send_msg(self.sock, K_GET, repr(key).encode())
            k, data, fds = _recv_msg(self.sock)
        except (OSError, EOFError, BrokerError):
            return None                                   # (a cache: the call that needs the broker will say that it is gone)
        if k != K_REPLY or not fds:
            for fd in fds:
                os.close(fd)
            return None
        try:
            size = struct.unpack("<Q", data[:8])[0]
            mm = mmap.mmap(fds[0], size, prot=mmap.PROT_READ)
            return WalkedReads.from_buffer(mm)             # (the arrays keep the mapping alive)
        except (OSError, ValueError):
            return None
        finally:
            for fd in fds:
                os.close(fd)

    def reads_flush(self):
        """forget every walked block (bench_stage.py: a timed stage must not find the blocks of the one before)"""
        self._request(K_FLUSH)

    def reads_put(self, key, wr):
        """leave a walked block with the broker (an anonymous shared-memory file; the broker only keeps the descriptor)"""
        size = wr.nbytes()
        fd = os.memfd_create("cutesv_amd_reads", 0)
        try:
            os.ftruncate(fd, size)
            wr.write_fd(fd)
            self._request(K_PUT, struct.pack("<Q", size) + repr(key).encode(), fds=(fd,))
        except (OSError, BrokerError):
            pass                                          # (a cache: failing to share a block loses nothing but time)
        finally:
            os.close(fd)

    # ---- the call
    def _ensure(self, need):
        if self.region is not None and self.region.size >= need:
            self.region.reset()
            return
        if self.region is not None:
            self.region.close()
        self.region = Region(max(need * 5 // 4, 4 << 20))
        rc = struct.unpack("<i", self._request(K_MAP, struct.pack("<Q", self.region.size), fds=(self.region.fd,))[:4])[0]
        if rc != _abi.OK:
            raise BrokerError("the broker could not map the shared region (%d bytes)" % self.region.size)

    @staticmethod
    def _compact(batch):
        """What travels for `batch`: (columns by csv_batch_in field, signature base, n_chrom).  A single-segment batch over a
        larger store (a worker on the genome's `.cols` files) sends the segment's rows and its chromosome's reads block only;
        the result's signature indices are moved back by `base` afterwards."""
        segs = batch.segments
        whole = [("seg", segs), ("a", batch.a), ("b", batch.b), ("read_id", batch.read_id), ("aux", batch.aux),
                 ("reads_off", batch.reads_off), ("r_start", batch.r_start), ("r_end", batch.r_end), ("r_primary", batch.r_primary),
                 ("r_id", batch.r_id), ("contig_len", batch.contig_len)]
        if len(segs) != 1 or batch.a is None or batch.contig_len is not None:
            return whole, 0, batch.n_chrom, batch.n_sig
        sg = segs.copy()
        lo, hi = int(sg[0]["sig_begin"]), int(sg[0]["sig_end"])
        one_chrom = batch.reads_off is None or batch.n_chrom == 1
        if lo == 0 and hi == batch.n_sig and one_chrom:
            return whole, 0, batch.n_chrom, batch.n_sig
        sg[0]["sig_begin"], sg[0]["sig_end"] = 0, hi - lo
        cols = [("seg", sg), ("a", batch.a[lo:hi]), ("b", batch.b[lo:hi]), ("read_id", batch.read_id[lo:hi]), ("aux", batch.aux[lo:hi])]
        n_chrom = batch.n_chrom
        if batch.reads_off is not None:
            c = int(sg[0]["chrom"])
            r0, r1 = int(batch.reads_off[c]), int(batch.reads_off[c + 1])
            sg[0]["chrom"] = 0
            n_chrom = 1
            cols += [("reads_off", np.array([0, r1 - r0], np.int64)), ("r_start", batch.r_start[r0:r1]), ("r_end", batch.r_end[r0:r1]),
                     ("r_primary", batch.r_primary[r0:r1]), ("r_id", batch.r_id[r0:r1])]
        return cols, lo, n_chrom, hi - lo

    def cluster_batch(self, batch, per_sig=False, cap_calls=None, cap_support=None, reuse=False, no_support=False, coord32=False, fields=None):
        """engine.Context.cluster_batch through the broker.  The returned arrays live in this client's shared region and
        are overwritten by its next call (the contract of reuse=True; resolve.run_batch consumes them at once)."""
        from .engine import CsvError
        cols, base, n_chrom, n = self._compact(batch)
        if per_sig and base:
            raise ValueError("per-signature outputs of a sliced batch are not offered through the broker")
        cap_calls = cap_calls or max(64, n // 16 + 16)
        cap_support = cap_support or max(64, n + 16)
        in_bytes = sum(_padded(v.nbytes) for _, v in cols if v is not None)
        for _ in range(2):
            out_bytes = 16 * _padded(8 * (cap_calls + 1)) + _padded(8 * cap_support) + (2 * _padded(4 * n) if per_sig else 0) + _padded(4 * max(1, len(batch.segments)))
            self._ensure(in_bytes + out_bytes)
            reg = self.region
            cin = _abi.BatchIn.from_buffer_copy(bytes(batch.c))
            cin.n_chrom, cin.n_sig = n_chrom, n
            cin.a_delta = cin.a_esc_row = cin.a_esc_val = cin.rows8 = None      # (the gap / interleaved forms of a whole store's columns do not travel with a task)
            cin.r_delta = cin.r_esc_row = cin.r_esc_val = cin.r_len16 = cin.l_esc_row = cin.l_esc_val = cin.r_idp = None
            cin.n_esc = cin.n_r_esc = cin.n_l_esc = 0
            cin.flags &= ~(_abi.IN_SIG_DELTA16 | _abi.IN_READS_DELTA16)
            by = dict(cols)
            # positions and lengths travel as int32 when they fit (a genome's coordinates do: CSV_IN_SIG_I32 / CSV_IN_READS_I32):
            # half the bytes through the region, the broker's staging columns and the link
            sig32 = _fits32(by.get("a"), by.get("b"))
            rd32 = _fits32(by.get("r_start"), by.get("r_end"))
            cin.flags = (cin.flags & ~(_abi.IN_SIG_I32 | _abi.IN_READS_I32)) | (_abi.IN_SIG_I32 if sig32 else 0) | (_abi.IN_READS_I32 if rd32 else 0)
            for name, v in cols:
                if v is not None:
                    narrow = (sig32 and name in ("a", "b")) or (rd32 and name in ("r_start", "r_end"))
                    setattr(cin, name, (reg.put(v, np.int32) if narrow else reg.put(v)).ctypes.data)
                    if name == "r_start":
                        cin.n_reads = len(v)
            res = _abi.HostResult(n, cap_calls, cap_support, per_sig=per_sig, n_seg=len(batch.segments), alloc=reg.alloc,
                                  narrow_support=bool(reuse), no_support=no_support, coord32=coord32, fields=fields, seg_alloc=reg.alloc)
            data = self._request(K_CALL, struct.pack("<Q", reg.base) + bytes(cin) + bytes(res.c))
            rc, _cc, _cs, res.c.n_calls, res.c.n_support, res.c.n_clusters = REPLY.unpack(data[:REPLY.size])
            self.calls += 1
            if rc == _abi.E_CAPACITY:
                cap_calls, cap_support = max(cap_calls, res.n_calls + 1), max(cap_support, res.n_support + 1)
                continue
            if rc != _abi.OK:
                raise CsvError(rc, data[REPLY.size:].decode("utf-8", "replace"))
            res.n_seg_used = len(batch.segments)
            if base:                                  # signature indices of the slice -> of the caller's columns
                t = res.trimmed()
                if t["support_sig"] is not None:
                    t["support_sig"] += base
                if t["seq_pick"] is not None:
                    sp = t["seq_pick"]
                    sp[sp >= 0] += base
            return res
        raise CsvError(_abi.E_CAPACITY, "capacity retry failed")


def _try_connect(name):
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        s.connect("\0" + name)
        return s
    except OSError:
        s.close()
        return None


def spawn(name, device, watch_pid, linger=None, log=None, prealloc=None):
    """start the broker process of `name` (a fresh interpreter: no HIP state is inherited); returns the Popen object.
    prealloc: "signatures,reads" - staging page-locked before the first request (resolve.warm_up passes it)"""
    import subprocess
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    cmd = [sys.executable, "-m", "cutesv_amd.broker", "--name", name, "--device", str(device), "--watch-pid", str(watch_pid)]
    if linger is not None:
        cmd += ["--linger", str(linger)]
    if prealloc:
        cmd += ["--prealloc", str(prealloc)]
    log = log or os.environ.get("CUTESV_AMD_BROKER_LOG")
    err = open(log, "ab") if log else None
    try:
        return subprocess.Popen(cmd, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=err, env=env,
                                start_new_session=True, close_fds=True)
    finally:
        if err is not None:
            err.close()


def _spawn_and_connect(name, device, watch_pid, timeout):
    import fcntl
    import tempfile
    lock_path = os.path.join(tempfile.gettempdir(), name + ".lock")
    with open(lock_path, "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)                # (the other workers of the pool queue here while the first one starts it)
        try:
            s = _try_connect(name)
            if s is not None:
                return s
            proc = spawn(name, device, watch_pid)
            t_end = time.monotonic() + timeout
            while time.monotonic() < t_end:
                s = _try_connect(name)                # (the broker listens BEFORE it creates its context: this returns early)
                if s is not None:
                    return s
                if proc.poll() is not None:
                    raise BrokerError("the GPU broker exited with code %s while starting (device %d); set CUTESV_AMD_BROKER_LOG "
                                      "to a file to see why. There is no CPU path." % (proc.returncode, device))
                time.sleep(0.005)
            raise BrokerError("the GPU broker did not come up within %.0f s" % timeout)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
            try:
                os.unlink(lock_path)
            except OSError:
                pass


# =============================================================================================== broker side
class _Conn:
    __slots__ = ("sock", "mm", "base", "size", "registered")

    def __init__(self, sock):
        self.sock, self.mm, self.base, self.size, self.registered = sock, None, 0, 0, False


class _Pending:
    __slots__ = ("conn", "cin", "cout", "delta")

    def __init__(self, conn, cin, cout, delta):
        self.conn, self.cin, self.cout, self.delta = conn, cin, cout, delta


def _rebase(st, names, delta, lo, hi):
    """pointers of the requester's address space -> ours; False if one points outside its region"""
    for f in names:
        p = getattr(st, f)
        if p:
            q = p + delta
            if not lo <= q <= hi:
                return False
            setattr(st, f, q)
    return True


def _view(addr, n, dtype):
    dtype = np.dtype(dtype)
    if n <= 0 or not addr:
        return np.zeros(0, dtype)
    return np.frombuffer((C.c_char * (n * dtype.itemsize)).from_address(addr), dtype=dtype, count=n)


class Broker:
    """The serving loop.  `call(handle, cin, cout) -> rc` and `last_error()` come from the engine: libcutesv_hip.so through an
    `engine.Context` in the product; the tests hand in the oracle's entry point to exercise the protocol without a GPU."""

    def __init__(self, name, device=0, watch_pid=0, linger=30.0, engine_factory=None, max_batch=64, prealloc=None):
        self.name, self.device, self.watch_pid, self.linger = name, int(device), int(watch_pid), float(linger)
        self.max_batch = int(os.environ.get("CUTESV_AMD_BROKER_BATCH", max_batch))
        self.listener = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self.listener.bind("\0" + name)               # (raises if a broker of this name is alive: exactly one per GPU and pool)
        self.listener.listen(512)
        self.conns = {}
        self.stats = dict(calls=0, batches=0, merged_calls=0, maps=0, max_batch=0, busy_s=0.0)
        self._stage = None
        self.prealloc = prealloc                      # (signatures, reads): staging and result arrays made before the first request
        self.blocks = {}                              # walked reads blocks the workers share: key -> (memfd, bytes)
        self.block_bytes = 0
        self.block_cap = int(os.environ.get("CUTESV_AMD_BROKER_BLOCKS_MB", "8192")) << 20
        self._pool = None
        self._copy_threads = int(os.environ.get("CUTESV_AMD_BROKER_COPY_THREADS", "8"))
        self._engine_factory = engine_factory
        self._engine = None
        self._ready_s = None

    # ---- engine
    def engine(self):
        if self._engine is None:
            t0 = time.perf_counter()
            self._engine = (self._engine_factory or _HipEngine)(self.device)
            self._ready_s = time.perf_counter() - t0
        return self._engine

    # ---- messages
    def _reply(self, conn, rc, out=None, text=""):
        vals = (out.cap_calls, out.cap_support, out.n_calls, out.n_support, out.n_clusters) if out is not None else (0, 0, 0, 0, 0)
        try:
            _send_msg(conn.sock, K_REPLY, REPLY.pack(int(rc), *[int(v) for v in vals]) + text.encode())
        except OSError:
            self._drop(conn)

    def _drop(self, conn):
        if self.conns.pop(conn.sock.fileno(), None) is None:
            return
        try:
            self.sel.unregister(conn.sock)
        except (KeyError, ValueError, AttributeError):
            pass
        self._unmap(conn)
        try:
            conn.sock.close()
        except OSError:
            pass

    def _unmap(self, conn):
        if conn.mm is not None:
            if conn.registered:
                self.engine().unregister(conn.base)
            conn.mm, conn.base, conn.size, conn.registered = None, 0, 0, False       # (unmapped with its last reference)

    def _on_map(self, conn, payload, fds):
        self._unmap(conn)
        size = struct.unpack("<Q", payload[:8])[0]
        try:
            conn.mm = mmap.mmap(fds[0], size)
            conn.base = C.addressof(C.c_char.from_buffer(conn.mm))
            conn.size = size
            # page-locked in place when the runtime agrees: the worker's columns then go up by DMA, straight from its pages
            conn.registered = self.engine().register(conn.base, size)
            self.stats["maps"] += 1
            rc = _abi.OK
        except (OSError, ValueError, IndexError):
            rc = _abi.E_NOMEM
        finally:
            for fd in fds:
                os.close(fd)
        try:
            _send_msg(conn.sock, K_REPLY, struct.pack("<i", rc))
        except OSError:
            self._drop(conn)

    def _parse_call(self, conn, payload):
        nin, nout = C.sizeof(_abi.BatchIn), C.sizeof(_abi.BatchOut)
        if len(payload) != 8 + nin + nout or conn.mm is None:
            return None
        their_base = struct.unpack("<Q", payload[:8])[0]
        cin = _abi.BatchIn.from_buffer_copy(payload[8:8 + nin])
        cout = _abi.BatchOut.from_buffer_copy(payload[8 + nin:])
        delta = conn.base - their_base
        lo, hi = conn.base, conn.base + conn.size
        if not (_rebase(cin, _IN_PTRS, delta, lo, hi) and _rebase(cout, _OUT_PTRS, delta, lo, hi)):
            return None
        return _Pending(conn, cin, cout, delta)

    # ---- running
    def _run_one(self, p):
        eng = self.engine()
        rc = eng.call(p.cin, p.cout)
        self._reply(p.conn, rc, p.cout, eng.last_error() if rc not in (_abi.OK, _abi.E_CAPACITY) else "")

    def _mergeable(self, p):
        """single-segment requests without per-signature outputs / TRA genotyping are laid side by side in one batch.  Cluster
        numbers are dense over a BATCH (the library counts chained clusters across segments), so a request that wants
        `call_cluster` runs alone; a merged request's `n_clusters` is -1 (not known per segment)."""
        c, o = p.cin, p.cout
        return (c.n_seg == 1 and not (c.flags & (_abi.IN_PER_SIG | _abi.IN_DEVICE_COLUMNS)) and not o.cluster_id and not o.allele_id
                and not o.call_cluster and not c.contig_len and c.n_sig > 0 and not (o.flags & _abi.OUT_COORD_I32))

    def _run(self, pend):
        t0 = time.perf_counter()
        self.stats["calls"] += len(pend)
        # Every single-segment request goes through the context's own page-locked staging columns and result arrays (a copy
        # at memory speed each way, then DMA) - alone or side by side with the requests that arrived with it.  Measured:
        # page-locking each worker's region in place instead (csv_host_register, 1-5 ms per region, serial in this process)
        # cost a 32-worker stage 180 ms of its 295.
        merge = [p for p in pend if self._mergeable(p)] if self.max_batch >= 1 else []
        ids = set(map(id, merge))
        for p in pend:
            if id(p) not in ids:
                self._run_one(p)
        for grp in self._groups(merge):
            self._run_merged(grp)
        self.stats["batches"] += 1
        self.stats["max_batch"] = max(self.stats["max_batch"], len(pend))
        self.stats["busy_s"] += time.perf_counter() - t0

    @staticmethod
    def _reads_block(c):
        """rows of the reads block a single-segment request's batch will carry (its segment's chromosome, if it genotypes)"""
        if not (c.n_reads and c.reads_off):
            return 0
        sg = _view(c.seg, 1, _abi.SEGMENT_DTYPE)[0]
        if not sg["genotype"]:
            return 0
        off = _view(c.reads_off, int(c.n_chrom) + 1, np.int64)
        return int(off[int(sg["chrom"]) + 1]) - int(off[int(sg["chrom"])])

    def _groups(self, merge):
        """the waiting requests cut into batches: at most max_batch side by side, and no more than the page-locked staging columns
        hold as they are - two batches cost one more call (~0.3 ms), growing a few hundred MB of page-locked memory in the middle of
        a stage costs a hundred (32 workers' reads blocks arriving at once)"""
        # (only a staging block that was sized on purpose - resolve.warm_up's prealloc - is a limit; one that grew out of the first
        # small request keeps growing, or nothing would ever be merged behind a broker started cold)
        st, cap = (self._stage if self.prealloc else None), max(1, self.max_batch)
        out, cur, n, r = [], [], 0, 0
        for p in merge:
            pn, pr = int(p.cin.n_sig), self._reads_block(p.cin)
            if cur and (len(cur) >= cap or (st is not None and (n + pn > st["n"] or r + pr > st["r"]))):
                out.append(cur)
                cur, n, r = [], 0, 0
            cur.append(p)
            n += pn
            r += pr
        if cur:
            out.append(cur)
        return out

    def _run_merged(self, grp):
        """k single-segment requests -> one csv_cluster_batch.  Request j becomes segment j and "chromosome" j of the batch (its
        reads block, in its own read-id space: ids only have to tell reads apart inside a chromosome)."""
        eng = self.engine()
        k = len(grp)
        t_0 = time.perf_counter()
        wide_sig = any(not (p.cin.flags & _abi.IN_SIG_I32) for p in grp)
        wide_rd = any(p.cin.n_reads and not (p.cin.flags & _abi.IN_READS_I32) for p in grp)
        n_tot = sum(p.cin.n_sig for p in grp)
        r_tot = sum(p.cin.n_reads for p in grp)
        st = self._staging(eng, n_tot, r_tot, k)
        a = st["a"].view(np.int64 if wide_sig else np.int32)[:n_tot]
        b = st["b"].view(np.int64 if wide_sig else np.int32)[:n_tot]
        rid, aux = st["read_id"][:n_tot], st["aux"][:n_tot]
        rs = st["r_start"].view(np.int64 if wide_rd else np.int32)[:r_tot]
        re_ = st["r_end"].view(np.int64 if wide_rd else np.int32)[:r_tot]
        rp, ri = st["r_primary"][:r_tot], st["r_id"][:r_tot]
        segs, roff = st["seg"][:k], st["reads_off"][:k + 1]
        so = ro = 0
        roff[0] = 0
        copies = []                                   # (dst, src) pairs: done below, the large ones on a few threads

        def cp(dst, src):
            copies.append((dst, src))
        for j, p in enumerate(grp):
            c = p.cin
            n = int(c.n_sig)
            sdt = np.int32 if c.flags & _abi.IN_SIG_I32 else np.int64
            cp(a[so:so + n], _view(c.a, n, sdt))
            cp(b[so:so + n], _view(c.b, n, sdt))
            cp(rid[so:so + n], _view(c.read_id, n, np.int32))
            cp(aux[so:so + n], _view(c.aux, n, np.int32))
            sg = _view(c.seg, 1, _abi.SEGMENT_DTYPE)[0]
            segs[j] = sg
            segs[j]["sig_begin"] = so + int(sg["sig_begin"])
            segs[j]["sig_end"] = so + int(sg["sig_end"])
            segs[j]["chrom"] = j
            nr = 0
            if c.n_reads and sg["genotype"] and c.reads_off:
                # the request's own reads table may hold several chromosomes: only the segment's block travels
                off = _view(c.reads_off, int(c.n_chrom) + 1, np.int64)
                lo_, hi_ = int(off[int(sg["chrom"])]), int(off[int(sg["chrom"]) + 1])
                nr = hi_ - lo_
                rdt = np.int32 if c.flags & _abi.IN_READS_I32 else np.int64
                cp(rs[ro:ro + nr], _view(c.r_start, int(c.n_reads), rdt)[lo_:hi_])
                cp(re_[ro:ro + nr], _view(c.r_end, int(c.n_reads), rdt)[lo_:hi_])
                cp(rp[ro:ro + nr], _view(c.r_primary, int(c.n_reads), np.uint8)[lo_:hi_])
                cp(ri[ro:ro + nr], _view(c.r_id, int(c.n_reads), np.int32)[lo_:hi_])
            so += n
            ro += nr
            roff[j + 1] = ro
        self._copy_all(copies)
        t_1 = time.perf_counter()
        any_reads = ro > 0
        cin = _abi.BatchIn(n_seg=k, n_chrom=k, seg=segs.ctypes.data, n_sig=n_tot, a=a.ctypes.data, b=b.ctypes.data,
                           read_id=rid.ctypes.data, aux=aux.ctypes.data,
                           reads_off=roff.ctypes.data if any_reads else None, n_reads=ro if any_reads else 0,
                           r_start=rs.ctypes.data if any_reads else None, r_end=re_.ctypes.data if any_reads else None,
                           r_primary=rp.ctypes.data if any_reads else None, r_id=ri.ctypes.data if any_reads else None,
                           flags=(0 if wide_sig else _abi.IN_SIG_I32) | (0 if (wide_rd or not any_reads) else _abi.IN_READS_I32))
        res = st.get("res")
        for _ in range(2):
            cap_c, cap_s = max(64, n_tot // 16 + 16), max(64, n_tot + 16)
            if res is None or res.cap_calls < cap_c or res.cap_support < cap_s or res.n_seg < k:
                res = st["res"] = _abi.HostResult(n_tot, max(cap_c, st.get("need_c", 0)), max(cap_s, st.get("need_s", 0)), n_seg=max(k, self.max_batch),
                                                  alloc=eng.alloc, narrow_support=True)
            rc = eng.call(cin, res.c)
            if rc == _abi.E_CAPACITY:
                st["need_c"], st["need_s"] = res.n_calls + 1, res.n_support + 1
                res = None
                continue
            break
        t_2 = time.perf_counter()
        self.stats["stage_in_s"] = self.stats.get("stage_in_s", 0.0) + (t_1 - t_0)
        self.stats["engine_s"] = self.stats.get("engine_s", 0.0) + (t_2 - t_1)
        if rc != _abi.OK:
            text = eng.last_error()
            for p in grp:                             # (a batch-level failure: every requester is told; one bad request cannot hide)
                self._reply(p.conn, rc, None, text)
            return
        if k > 1:
            self.stats["merged_calls"] += k
        t = res.trimmed()
        cut = np.searchsorted(t["call_seg"], np.arange(k + 1))
        soff = t["support_off"]
        seg_begin = segs["sig_begin"]
        for j, p in enumerate(grp):
            o = p.cout
            lo_, hi_ = int(cut[j]), int(cut[j + 1])
            nc = hi_ - lo_
            s0, s1 = int(soff[lo_]), int(soff[hi_])
            ns = s1 - s0
            o.n_calls, o.n_support = nc, (0 if o.flags & _abi.OUT_NO_SUPPORT_LIST else ns)
            o.n_clusters = int(res.n_clusters) if k == 1 else -1
            if nc > o.cap_calls or (not (o.flags & _abi.OUT_NO_SUPPORT_LIST) and ns > o.cap_support):
                self._reply(p.conn, _abi.E_CAPACITY, o)
                continue
            base_sig = int(seg_begin[j]) - int(_view(p.cin.seg, 1, _abi.SEGMENT_DTYPE)[0]["sig_begin"])
            for name, dt, cap in _abi._OUT_ARRAYS:
                dst = getattr(o, name)
                if cap == "calls" and dst:
                    src = t[name][lo_:hi_]
                    if name == "call_seg":
                        src = np.zeros(nc, np.int32)
                    elif name == "seq_pick":
                        src = np.where(src >= 0, src - base_sig, src)
                    _view(dst, nc, dt)[:] = src
            if not (o.flags & _abi.OUT_NO_SUPPORT_LIST):
                if o.support_off:
                    _view(o.support_off, nc + 1, np.int64)[:] = soff[lo_:hi_ + 1] - s0
                ss = t["support_sig"][s0:s1].astype(np.int64) - base_sig
                if o.support_sig32:
                    _view(o.support_sig32, ns, np.int32)[:] = ss
                elif o.support_sig:
                    _view(o.support_sig, ns, np.int64)[:] = ss
            if o.seg_status:
                _view(o.seg_status, 1, np.int32)[0] = t["seg_status"][j]
            self._reply(p.conn, _abi.OK, o)
        self.stats["slice_out_s"] = self.stats.get("slice_out_s", 0.0) + (time.perf_counter() - t_2)

    def _copy_all(self, copies):
        """shared region -> staging.  numpy's copy loops release the GIL: a merged batch's large columns (a chromosome's reads
        table is megabytes) are copied by a few threads at once instead of one after the other"""
        big = [c for c in copies if c[0].nbytes >= (1 << 19)]
        if len(big) >= 2 and self._copy_threads > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self._copy_threads)
            futs = [self._pool.submit(np.copyto, d, s_) for d, s_ in big]
            for d, s_ in copies:
                if d.nbytes < (1 << 19):
                    np.copyto(d, s_)
            for f in futs:
                f.result()
        else:
            for d, s_ in copies:
                np.copyto(d, s_)

    def _staging(self, eng, n, r, k):
        st = self._stage
        if st is None or st["n"] < n or st["r"] < r or st["k"] < k:
            self.stats["stage_grows"] = self.stats.get("stage_grows", 0) + 1
            n2, r2, k2 = max(n * 5 // 4, 1 << 16), max(r * 5 // 4, 1 << 16), max(k, self.max_batch)
            st = self._stage = dict(n=n2, r=r2, k=k2,
                                    a=eng.alloc(n2, np.int64), b=eng.alloc(n2, np.int64), read_id=eng.alloc(n2, np.int32), aux=eng.alloc(n2, np.int32),
                                    r_start=eng.alloc(r2, np.int64), r_end=eng.alloc(r2, np.int64), r_primary=eng.alloc(r2, np.uint8), r_id=eng.alloc(r2, np.int32),
                                    seg=np.zeros(k2, _abi.SEGMENT_DTYPE), reads_off=np.zeros(k2 + 1, np.int64))
        return st

    # ---- the loop
    def _handle(self, s, pend):
        """one readable socket: a new worker, or one message of a worker; True = a shutdown request"""
        import json
        if s is self.listener:
            try:
                c, _ = self.listener.accept()
            except OSError:
                return False
            if _peer_uid(c) != os.getuid():
                c.close()
                return False
            c.settimeout(10.0)                        # (a message is sent in one piece: a peer that stalls mid-message is dropped, not waited for)
            self.conns[c.fileno()] = _Conn(c)
            self.sel.register(c, selectors.EVENT_READ)
            self._served = True
            return False
        conn = self.conns.get(s.fileno())
        if conn is None:
            return False
        try:
            kind, payload, fds = _recv_msg(s)
        except (EOFError, OSError, BrokerError):
            self._drop(conn)
            return False
        if kind == K_MAP:
            self._on_map(conn, payload, fds)
        elif kind == K_CALL:
            p = self._parse_call(conn, payload)
            if p is None:
                self._reply(conn, _abi.E_INVALID, None, "broker: a request whose pointers do not lie in its shared region")
            else:
                pend.append(p)
        elif kind == K_PUT:
            try:
                size = struct.unpack("<Q", payload[:8])[0]
                key = payload[8:]
                if fds and key not in self.blocks and self.block_bytes + size <= self.block_cap:
                    self.blocks[key] = (fds[0], size)
                    self.block_bytes += size
                    fds = fds[1:]
                    self.stats["blocks"] = len(self.blocks)
            finally:
                for fd in fds:
                    os.close(fd)
            self._reply(conn, _abi.OK)
        elif kind == K_FLUSH:
            self._forget_blocks()
            self._reply(conn, _abi.OK)
        elif kind == K_GET:
            ent = self.blocks.get(payload)
            try:
                if ent is None:
                    _send_msg(s, K_REPLY, b"")
                else:
                    self.stats["block_hits"] = self.stats.get("block_hits", 0) + 1
                    _send_msg(s, K_REPLY, struct.pack("<Q", ent[1]), fds=(ent[0],))
            except OSError:
                self._drop(conn)
        elif kind == K_INFO:
            d = dict(self.stats, pid=os.getpid(), device=self.device, name=self.name, clients=len(self.conns),
                     engine_start_s=self._ready_s, **self.engine().describe())
            try:
                _send_msg(s, K_REPLY, json.dumps(d).encode())
            except OSError:
                self._drop(conn)
        elif kind == K_SHUTDOWN:
            try:
                _send_msg(s, K_REPLY, b"")
            except OSError:
                pass
            return True
        else:
            self._drop(conn)
        return False

    def serve(self):
        self.sel = selectors.DefaultSelector()
        self.sel.register(self.listener, selectors.EVENT_READ)
        self._served = False
        eng = self.engine()                           # (clients are already queueing on the listening socket)
        if self.prealloc:
            # a broker started ahead of the stage (resolve.warm_up) also page-locks its staging columns and result arrays now:
            # grown on demand they cost the FIRST batches of a stage 50-100 ms (hipHostMalloc of a few hundred MB), which a
            # pool of 32 workers - all of whose first requests arrive together - pays in full
            try:
                n, r = int(self.prealloc[0]), int(self.prealloc[1])
                st = self._staging(eng, n, r, self.max_batch)
                st["res"] = _abi.HostResult(n, max(64, n // 16 + 16), max(64, n + 16), n_seg=self.max_batch, alloc=eng.alloc, narrow_support=True)
            except Exception as e:                      # noqa: BLE001  (an optimisation: the broker serves without it)
                sys.stderr.write("cutesv_amd.broker: staging not pre-allocated (%r)\n" % (e,))
        idle_since = time.monotonic()
        # CUTESV_AMD_BROKER_GATHER_MS: after a request arrives, wait this long for the other workers' requests before
        # launching (0: take what is there - the requests that piled up behind the previous batch are merged anyway)
        gather = float(os.environ.get("CUTESV_AMD_BROKER_GATHER_MS", "0")) * 1e-3
        while True:
            ready = [key.fileobj for key, _ in self.sel.select(0.25)]
            if not ready:
                if self.watch_pid and not _alive(self.watch_pid):
                    return "owner gone"
                if not self.conns and time.monotonic() - idle_since > (self.linger if self._served else max(self.linger, 120.0)):
                    return "idle"
                continue
            pend = []
            for s in ready:
                if self._handle(s, pend):
                    return "shutdown"
            if pend and gather > 0:
                t_end = time.monotonic() + gather
                while len(pend) < len(self.conns) and time.monotonic() < t_end:
                    for key, _ in self.sel.select(max(0.0, t_end - time.monotonic())):
                        if self._handle(key.fileobj, pend):
                            return "shutdown"
            if pend:
                self._run(pend)
            if self.conns:
                idle_since = time.monotonic()

    def _forget_blocks(self):
        for fd, _ in self.blocks.values():
            try:
                os.close(fd)
            except OSError:
                pass
        self.blocks, self.block_bytes = {}, 0
        self.stats["blocks"] = 0

    def close(self):
        for c in list(self.conns.values()):
            self._drop(c)
        self._forget_blocks()
        self.listener.close()
        if self._engine is not None:
            self._engine.close()


def _alive(pid):
    try:
        os.kill(pid, 0)
        return True
    except ProcessLookupError:
        return False
    except PermissionError:
        return True


def _peer_uid(sock):
    cred = sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
    return struct.unpack("3i", cred)[1]


class _HipEngine:
    """libcutesv_hip.so behind the broker: one engine.Context, page-locked staging from csv_host_alloc"""

    def __init__(self, device):
        from . import engine
        from ._lib import lib
        self._engine_mod, self._lib = engine, lib()
        self.ctx = engine.Context(device)

    def call(self, cin, cout):
        return self._lib.csv_cluster_batch(self.ctx._h, C.byref(cin), C.byref(cout))

    def last_error(self):
        return (self._lib.csv_last_error(self.ctx._h) or b"").decode()

    def alloc(self, shape, dtype):
        return self._engine_mod.pinned_empty(shape, dtype)

    def register(self, addr, size):
        if os.environ.get("CUTESV_AMD_BROKER_REGISTER", "0") != "1":     # (off by default: see Broker._run)
            return False
        return self._lib.csv_host_register(C.c_void_p(addr), int(size)) == _abi.OK

    def unregister(self, addr):
        self._lib.csv_host_unregister(C.c_void_p(addr))

    def describe(self):
        bus, ncu = self._engine_mod.device_info(self.ctx.device)
        return dict(bus=bus, compute_units=ncu, engine="libcutesv_hip.so")

    def close(self):
        self.ctx.close()


def _reserve_fd_table(n=4096):
    """Grow this process's descriptor table ONCE, now, while it has one thread.  The kernel enlarges the table in powers of two and, in
    a process with several threads (the HIP runtime's, the copy pool's), waits for an RCU grace period each time: a pool of 32
    workers connecting for the first time - 32 sockets, 32 region descriptors - crossed 64 / 128 / 256 descriptors inside the stage
    and the broker sat 150-180 ms in accept / recvmsg (measured: one K_MAP message 176 ms, a 32-worker stage 263 instead of 151 ms;
    the next stage, its table already large, was fast)."""
    if os.environ.get("CUTESV_AMD_BROKER_FD_TABLE", "1") == "0":           # (diagnostic: the table grows on demand again)
        return
    try:
        import resource
        soft, _hard = resource.getrlimit(resource.RLIMIT_NOFILE)
        hi = min(int(n), int(soft) - 1)
        if hi > 64:
            os.dup2(0, hi, inheritable=False)
            os.close(hi)
    except (OSError, ValueError, ImportError):
        pass


def main(argv=None, engine_factory=None):
    _reserve_fd_table()
    import argparse
    ap = argparse.ArgumentParser(description="cutesv_amd GPU broker: one HIP context per GPU for a pool of cuteSV workers")
    ap.add_argument("--name", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--watch-pid", type=int, default=0)
    ap.add_argument("--linger", type=float, default=float(os.environ.get("CUTESV_AMD_BROKER_LINGER", "30")))
    ap.add_argument("--prealloc", default="", help="signatures,reads: page-lock staging for batches of that size before the first request")
    a = ap.parse_args(argv)
    pre = tuple(int(x) for x in a.prealloc.split(",")) if a.prealloc else None
    try:
        b = Broker(a.name, a.device, a.watch_pid, a.linger, engine_factory=engine_factory, prealloc=pre if pre and len(pre) == 2 else None)
    except OSError as e:                              # (somebody else bound the name between our starter's probe and now: fine)
        sys.stderr.write("cutesv_amd.broker: %s is taken (%s)\n" % (a.name, e))
        return 0
    try:
        why = b.serve()
        sys.stderr.write("cutesv_amd.broker %s: exit (%s) after %d calls in %d batches (%d merged), %.3f s busy\n"
                         % (a.name, why, b.stats["calls"], b.stats["batches"], b.stats["merged_calls"], b.stats["busy_s"]))
    finally:
        b.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
