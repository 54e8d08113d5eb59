"""Multi-GPU sharding of the MSM (one process per GPU, torch.distributed; backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

Inside mh_marlin_prove every rank recodes every scalar but sorts, accumulates and reduces only the digits that fall
into ITS partitions of the shared bucket set: partition v (2^11 consecutive buckets) belongs to rank v mod G --
interleaved, because the low buckets are the heavier ones (bucket-range sharding).  A window table with
fewer partitions than ranks (small SRS), a short vector or the skew fallback make a rank compute a group in full
instead; the payload carries a share / whole flag per job and a whole result takes precedence.  `msm_sharded` below
is the plain point-sharded variant for callers that hold only a slice of the bases.  The per-rank partial results
(one Jacobian point, 144 bytes) are exchanged with ONE all_gather per batch of MSMs and summed on every rank, so
all ranks derive the same commitments and the same Fiat-Shamir challenges.  Elliptic-curve addition is not an RCCL reduction operator,
hence all_gather + local adds instead of all_reduce.
"""
import ctypes as C
import numpy as np
from . import _lib


def shard_range(n, rank, world):
    """contiguous, balanced split of n (scalar, base) pairs."""
    return (n * rank) // world, (n * (rank + 1)) // world


def g1_sum(points_xyz):
    """sum of Jacobian points ((k,18) uint64 Montgomery) on the host."""
    _lib.load()
    L = 3 * _lib.FQ_LIMBS
    pts = np.ascontiguousarray(points_xyz, dtype=np.uint64).reshape(-1, L)
    out = np.zeros(L, dtype=np.uint64)
    _lib.check(_lib.load().mh_g1_sum(pts.ctypes.data, pts.shape[0], out.ctypes.data), "mh_g1_sum")
    return out


def allgather_partials(partials_xyz, dist, device=None):
    """partials_xyz: (m,18) uint64 = this rank's partial result of m MSMs.  Returns (world, m, 18)."""
    import torch
    _lib.load()
    local = np.ascontiguousarray(partials_xyz, dtype=np.uint64).reshape(-1, 3 * _lib.FQ_LIMBS)
    t = torch.from_numpy(local.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in out])


def combine_partials(gathered):
    """(world, m, 18) -> (m, 18): per MSM, the sum over ranks."""
    g = np.asarray(gathered, dtype=np.uint64)
    return np.stack([g1_sum(g[:, j, :]) for j in range(g.shape[1])])


def msm_sharded(bases_shard, d_scalars_shard, counts, dist, device=None, msm_dev=None):
    """Run this rank's share of a batch of MSMs and combine across ranks.
    counts[j] = number of local pairs of MSM j (prefix of the local shard)."""
    from .api import msm_dev as _msm_dev
    f = msm_dev or _msm_dev
    partials = np.stack([f(bases_shard, d_scalars_shard, int(cnt)) for cnt in counts])
    if dist is None or dist.get_world_size() == 1:
        return partials
    return combine_partials(allgather_partials(partials, dist, device))


# ---- sharded Marlin::prove: register torch.distributed's all_gather as the library's exchange step ----
_ALLGATHER_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
_keepalive = {}


def enable_sharded_prove(dist, device=None):
    """After this call mh_marlin_prove sorts, accumulates and reduces only this rank's partitions of every MSM's bucket set
    and combines the partial points across ranks (mh_marlin_set_shard).  The exchange runs four times per proof inside the
    timed region, so the callback does as little as Python allows: the payload (<= 1.4 KB) is copied into a persistent
    (pinned, when the transport is RCCL) staging tensor, ONE all_gather_into_tensor moves it, and the result is copied
    straight into the library's receive buffer -- no per-call allocation, no list of tensors, no numpy round trips."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    st = {"n": 0}

    def _buffers(nbytes):
        # grow-only: the four exchanges of a proof carry different job counts, and a pinned allocation per size change would
        # cost more than the exchange; the calls use views of the size they need
        if st["n"] < nbytes:
            cap = max(nbytes, 4096)
            pin = device is not None
            st["send_H"] = torch.empty(cap, dtype=torch.uint8, pin_memory=pin)
            st["recv_H"] = torch.empty(cap * world, dtype=torch.uint8, pin_memory=pin)
            if device is not None:
                st["send_D"] = torch.empty(cap, dtype=torch.uint8, device=device)
                st["recv_D"] = torch.empty(cap * world, dtype=torch.uint8, device=device)
            st["n"] = cap
            st["views"] = {}
        v = st["views"].get(nbytes)
        if v is None:
            v = {"send_h": st["send_H"][:nbytes], "recv_h": st["recv_H"][:nbytes * world]}
            if device is not None:
                v["send_d"], v["recv_d"] = st["send_D"][:nbytes], st["recv_D"][:nbytes * world]
            st["views"][nbytes] = v
        return v

    # the flat variant is probed ONCE, here, by capability (gloo and RCCL both have it in this torch) -- a RuntimeError inside the
    # callback is then a transport failure and returns -1 instead of being retried as a second collective the peers do not issue
    flat = hasattr(dist, "all_gather_into_tensor")

    def _gather(dst, src):
        if flat:
            dist.all_gather_into_tensor(dst, src)
        else:
            dist.all_gather(list(dst.view(world, -1).unbind(0)), src)

    def _cb(send, nbytes, recv, _user):
        try:
            b = _buffers(nbytes)
            C.memmove(b["send_h"].data_ptr(), send, nbytes)
            if device is not None:
                b["send_d"].copy_(b["send_h"], non_blocking=True)
                _gather(b["recv_d"], b["send_d"])
                b["recv_h"].copy_(b["recv_d"])                  # synchronises the transport's stream with the host
            else:
                _gather(b["recv_h"], b["send_h"])
            C.memmove(recv, b["recv_h"].data_ptr(), nbytes * world)
            return 0
        except Exception as e:      # never unwind into C
            import sys
            print("all_gather callback failed:", e, file=sys.stderr)
            return -1
    cb = _ALLGATHER_T(_cb)
    _keepalive["cb"] = cb
    _lib.check(_lib.load().mh_marlin_set_shard(rank, world, C.cast(cb, C.c_void_p), None), "mh_marlin_set_shard")


def enable_native_rccl(dist=None, sliced=True):
    """The native transport (marlin_amd/csrc/rccl_native.h): the library creates its OWN RCCL communicator and issues the
    all-gathers / all-to-alls of a sharded proof from C++ on its own stream -- no Python in the exchange path.  `dist` (an
    initialised torch.distributed, any backend) is used ONCE, to hand rank 0's ncclUniqueId to the other ranks; dist = None
    makes a communicator of one rank (tests on a one-GPU box).  Every rank first probes that it can reach librccl at all and the
    ranks agree on the outcome before anyone enters the collective ncclCommInitRank, so a box without RCCL makes this return
    False everywhere instead of hanging.  Returns True when the native transport is active on every rank."""
    # torch is touched only when a process group is handed in: a process that has loaded libmarlin_hip.so WITHOUT torch runs on
    # /opt/rocm's HIP runtime, and importing torch afterwards would map torch's bundled copy of libamdhip64 / libhsa-runtime64 /
    # librccl next to it (its libraries ask for "libamdhip64.so", not the SONAME) -- two HIP runtimes in one process, the second of
    # which finds no device.  A torch process imports torch FIRST (bench.py, the tests); then libmarlin_hip.so binds to torch's copy.
    lib = _lib.load()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    ident = np.zeros(128, dtype=np.uint8)
    ok = lib.mh_rccl_unique_id(ident.ctypes.data) == 0            # every rank draws one (the probe); rank 0's is the one used
    if dist is not None and world > 1:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(int(flag.item()))
        if ok:
            t = torch.from_numpy(ident).to(dev)
            dist.broadcast(t, src=0)
            ident = np.ascontiguousarray(t.cpu().numpy())
    if not ok:
        return False
    _lib.check(lib.mh_marlin_set_rccl(rank, world, ident.ctypes.data), "mh_marlin_set_rccl")
    if not sliced:
        _lib.check(lib.mh_marlin_rccl_sliced(0), "mh_marlin_rccl_sliced")
    return True


def enable_native_rccl_solo(rank, world, sliced=True):
    """MEASUREMENT ONLY (bench.py --simulate-rank R/G --transport native): the native transport as rank `rank` of `world` without
    peers, over the stand-in for librccl in its solo mode (tests/mock_rccl/: MH_RCCL_LIB must point at it and MH_MOCK_RCCL_SOLO=1
    be set before the first call): every collective is a stream-ordered local copy issued from C++ -- what the native transport
    costs one rank's host, with no interpreter in the path.  Proofs made this way are not valid."""
    lib = _lib.load()
    ident = np.zeros(128, dtype=np.uint8)
    _lib.check(lib.mh_rccl_unique_id(ident.ctypes.data), "mh_rccl_unique_id")
    _lib.check(lib.mh_marlin_set_rccl(int(rank), int(world), ident.ctypes.data), "mh_marlin_set_rccl")
    if not sliced:
        _lib.check(lib.mh_marlin_rccl_sliced(0), "mh_marlin_rccl_sliced")


def native_rccl_info():
    """{"active", "allgather_host", "alltoall", "allgather_dev", "bytes_sent", "librccl"} of the native transport."""
    lib = _lib.load()
    info = (C.c_uint64 * 4)()
    path = C.create_string_buffer(512)
    active = lib.mh_marlin_rccl_info(info, path, 512)
    return {"active": bool(active), "allgather_host": int(info[0]), "alltoall": int(info[1]), "allgather_dev": int(info[2]),
            "bytes_sent": int(info[3]), "librccl": path.value.decode()}


def exchange_stats(reset=False):
    """(number of exchanges, host wall-clock ms inside them) since the last reset, whatever the transport."""
    calls, ms = C.c_uint64(), C.c_double()
    _lib.check(_lib.load().mh_marlin_exchange_stats(C.byref(calls), C.byref(ms), 1 if reset else 0), "mh_marlin_exchange_stats")
    return int(calls.value), float(ms.value)


def selftest_allgather(dist=None):
    """One all-gather of a rank-dependent payload through whatever transport is registered, checked on every rank."""
    lib = _lib.load()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    n = 600
    send = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(rank + 1)).view(np.uint8)
    recv = np.zeros(send.size * world, dtype=np.uint8)
    if lib.mh_marlin_probe_allgather(send.ctypes.data, send.size, recv.ctypes.data) != 0:
        return False
    want = np.concatenate([(np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(g + 1)).view(np.uint8) for g in range(world)])
    return bool(np.array_equal(recv, want))


def enable_simulated_shard(rank, world):
    """MEASUREMENT ONLY: make this process behave like rank `rank` of `world` without any peers -- the exchange is
    replaced by a local copy of this rank's own partial points into every slot, so the proof bytes are NOT valid, but
    every kernel the rank would run (its bucket range of every MSM, the replicated AHP rounds) runs and can be timed
    on a one-GPU box.  Used for the per-rank projection in DESIGN.md 8 (bench.py --simulate-rank)."""
    def _cb(send, nbytes, recv, _user):
        src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
        dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
        for g in range(world):
            dst[g * nbytes:(g + 1) * nbytes] = src
        return 0
    cb = _ALLGATHER_T(_cb)
    _keepalive["cb"] = cb
    _lib.check(_lib.load().mh_marlin_set_shard(rank, world, C.cast(cb, C.c_void_p), None), "mh_marlin_set_shard")


def disable_sharded_prove():
    _lib.check(_lib.load().mh_marlin_rccl_destroy(), "mh_marlin_rccl_destroy")
    _lib.check(_lib.load().mh_marlin_set_shard(0, 1, None, None), "mh_marlin_set_shard")
    _lib.check(_lib.load().mh_marlin_set_alltoall(None, None), "mh_marlin_set_alltoall")
    _keepalive.clear()


# ---- slice-sharded building blocks: the all-to-all of the distributed transform (mh_ntt_dist_dev) ----
_ALLTOALL_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)


def use_torch_stream(device):
    """Makes the library run on a stream torch knows (mh_set_stream): collectives issued under `torch.cuda.stream(s)` are then
    ordered with the library's kernels on the device, without host synchronisation.  Returns the stream (kept alive here)."""
    import torch
    s = torch.cuda.Stream(device=device)
    _lib.check(_lib.load().mh_set_stream(C.c_void_p(s.cuda_stream)), "mh_set_stream")
    _keepalive["stream"] = s
    return s


def enable_alltoall(dist, device=None, stream=None, allgather_dev=True):
    """Registers torch.distributed.all_to_all_single as the exchange of mh_ntt_dist_dev.  The library hands over DEVICE
    pointers of its own buffers.  RCCL (`device` given): torch wraps those buffers without a copy (__cuda_array_interface__) and
    the collective runs on them directly; if the wrap is refused, persistent device tensors are filled and drained with
    device-to-device copies -- either way the payload never touches the host.  With `stream` = the stream the library runs on
    (use_torch_stream) the collective is stream-ordered: RCCL's stream waits for the library's kernels and the library's
    next kernel waits for RCCL, the host waits for neither.  gloo (CPU tests): staged through host tensors."""
    import torch
    world = dist.get_world_size()
    lib = _lib.load()
    st = {"n": 0, "zero_copy": device is not None, "views": {}, "stream_ordered": device is not None and stream is not None, "calls": 0}

    class _DevMem:          # the library's device buffer as a __cuda_array_interface__ object: torch wraps it without a copy
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    def _view(ptr, nbytes):
        key = (int(ptr), int(nbytes))
        t = st["views"].get(key)
        if t is None:
            if len(st["views"]) > 64:
                st["views"].clear()
            t = torch.as_tensor(_DevMem(ptr, nbytes), device=device)
            if t.data_ptr() != int(ptr) or t.numel() != nbytes:
                raise RuntimeError("device view is not zero-copy")
            st["views"][key] = t
        return t

    def _cb(d_send, bytes_per_peer, d_recv, _user):
        try:
            total = bytes_per_peer * world
            if device is not None and st["zero_copy"]:
                # RCCL straight on the library's buffers: no staging copies; the library's stream is drained before the
                # collective reads, the collective before the library's next kernel
                try:
                    send, recv = _view(d_send, total), _view(d_recv, total)
                except Exception as e:
                    import sys
                    print("all_to_all: zero-copy views unavailable (%s); staging through torch tensors" % e, file=sys.stderr)
                    st["zero_copy"] = False
                    st["stream_ordered"] = False
                    lib.mh_marlin_set_alltoall_mode(0)
                else:
                    st["calls"] += 1
                    if st["stream_ordered"]:
                        try:
                            with torch.cuda.stream(stream):
                                dist.all_to_all_single(recv, send, async_op=True).wait()     # wait() = the stream waits, not the host
                            return 0
                        except Exception as e:
                            import sys
                            print("all_to_all: stream-ordered collective refused (%s); synchronising around it" % e, file=sys.stderr)
                            st["stream_ordered"] = False
                            lib.mh_marlin_set_alltoall_mode(0)
                    _lib.check(lib.mh_synchronize(), "sync")
                    dist.all_to_all_single(recv, send)
                    torch.cuda.synchronize(device)
                    return 0
            if st["n"] < total:                         # grow-only staging (exchanges of different sizes alternate)
                dev = device if device is not None else "cpu"
                st["send"] = torch.empty(total, dtype=torch.uint8, device=dev)
                st["recv"] = torch.empty(total, dtype=torch.uint8, device=dev)
                st["n"] = total
            send, recv = st["send"][:total], st["recv"][:total]
            if device is not None:
                _lib.check(lib.mh_memcpy_d2d(send.data_ptr(), d_send, total), "d2d")
                _lib.check(lib.mh_synchronize(), "sync")
                dist.all_to_all_single(recv, send)
                torch.cuda.synchronize(device)
                _lib.check(lib.mh_memcpy_d2d(d_recv, recv.data_ptr(), total), "d2d")
                _lib.check(lib.mh_synchronize(), "sync")
            else:
                _lib.check(lib.mh_memcpy_d2h(send.data_ptr(), d_send, total), "d2h")
                dist.all_to_all_single(recv, send)
                _lib.check(lib.mh_memcpy_h2d(d_recv, recv.data_ptr(), total), "h2d")
            return 0
        except Exception as e:      # never unwind into C
            import sys
            print("all_to_all callback failed:", e, file=sys.stderr)
            return -1
    def _ag(d_send, nbytes, d_recv, _user):
        """mh_allgather_dev_fn: the round polynomials of the sliced sections as ONE all_gather_into_tensor (each rank sends its
        chunk once; through the all-to-all it would send `world` copies)."""
        try:
            total = nbytes * world
            if device is not None and st["zero_copy"]:
                send, recv = _view(d_send, nbytes), _view(d_recv, total)
                st["calls"] += 1
                if st["stream_ordered"]:
                    with torch.cuda.stream(stream):
                        dist.all_gather_into_tensor(recv, send, async_op=True).wait()
                    return 0
                _lib.check(lib.mh_synchronize(), "sync")
                dist.all_gather_into_tensor(recv, send)
                torch.cuda.synchronize(device)
                return 0
            dev = device if device is not None else "cpu"
            if st.get("ag_n", 0) < total:
                st["ag_send"] = torch.empty(total, dtype=torch.uint8, device=dev)
                st["ag_recv"] = torch.empty(total, dtype=torch.uint8, device=dev)
                st["ag_n"] = total
            send, recv = st["ag_send"][:nbytes], st["ag_recv"][:total]
            if device is not None:
                _lib.check(lib.mh_memcpy_d2d(send.data_ptr(), d_send, nbytes), "d2d")
                _lib.check(lib.mh_synchronize(), "sync")
                dist.all_gather_into_tensor(recv, send)
                torch.cuda.synchronize(device)
                _lib.check(lib.mh_memcpy_d2d(d_recv, recv.data_ptr(), total), "d2d")
                _lib.check(lib.mh_synchronize(), "sync")
            else:
                _lib.check(lib.mh_memcpy_d2h(send.data_ptr(), d_send, nbytes), "d2h")
                dist.all_gather_into_tensor(recv, send)
                _lib.check(lib.mh_memcpy_h2d(d_recv, recv.data_ptr(), total), "h2d")
            return 0
        except Exception as e:      # never unwind into C
            import sys
            print("device all_gather callback failed:", e, file=sys.stderr)
            return -1
    cb = _ALLTOALL_T(_cb)
    _keepalive["a2a"] = cb
    _keepalive["a2a_state"] = st
    _lib.check(lib.mh_marlin_set_alltoall(C.cast(cb, C.c_void_p), None), "mh_marlin_set_alltoall")
    if allgather_dev:
        ag = _ALLTOALL_T(_ag)
        _keepalive["ag_dev"] = ag
        _lib.check(lib.mh_marlin_set_allgather_dev(C.cast(ag, C.c_void_p), None), "mh_marlin_set_allgather_dev")
    if st["stream_ordered"]:        # the library then calls the exchange without draining its stream first
        _lib.check(lib.mh_marlin_set_alltoall_mode(1), "mh_marlin_set_alltoall_mode")


def enable_simulated_alltoall(rank, world, stream_ordered=True):
    """MEASUREMENT ONLY (like enable_simulated_shard): this process acts as rank `rank` of `world` for mh_ntt_dist_dev and the
    sliced rounds of the prover without any peers.  The exchange copies this rank's own chunk into its slot; the slots of the
    absent peers hold pseudo-random field elements, written once per receive buffer -- so that what the prover then feeds to
    its MSMs looks like coefficient vectors (copies of the own slice in every slot would make every digit repeat `world`
    times and unbalance the bucket accumulation, which is not what a real run sees).  The values are meaningless and the
    proofs invalid; the kernels and the bytes each rank would move are the real ones (tools/ntt_dist_bench.py,
    bench.py --simulate-rank)."""
    lib = _lib.load()
    filled = {}

    def _cb(d_send, bytes_per_peer, d_recv, _user):
        total = bytes_per_peer * world
        if filled.get(d_recv, 0) < total:                     # once per receive buffer (and again only if a larger exchange uses it)
            rnd = np.random.default_rng(bytes_per_peer & 0xffff).integers(0, 1 << 63, size=(total // 32, 4), dtype=np.uint64)
            rnd[:, 3] &= np.uint64((1 << 60) - 1)
            if lib.mh_memcpy_h2d(d_recv, rnd.ctypes.data, total):
                return -1
            filled[d_recv] = total
        off = rank * bytes_per_peer
        # the copy is enqueued on the library's stream: like the RCCL transport of enable_alltoall(stream=...) it is ordered with
        # the library's kernels on the device, and (stream_ordered) nobody synchronises the host around it
        return lib.mh_memcpy_d2d(d_recv + off, d_send + off, bytes_per_peer) or (0 if stream_ordered else lib.mh_synchronize())
    cb = _ALLTOALL_T(_cb)
    _keepalive["a2a"] = cb
    _lib.check(lib.mh_marlin_set_alltoall(C.cast(cb, C.c_void_p), None), "mh_marlin_set_alltoall")
    if stream_ordered:
        _lib.check(lib.mh_marlin_set_alltoall_mode(1), "mh_marlin_set_alltoall_mode")
    enable_simulated_shard(rank, world)


def selftest_alltoall(dist, log_n=12):
    """One distributed transform of 2^log_n points checked against the same transform on this GPU alone (mh_ntt): True when
    this rank's block is bit-identical.  bench.py runs it once after registering the exchange, so that a transport problem
    turns the sliced rounds off (the replicated rounds need no all-to-all) instead of producing a wrong proof."""
    from .api import DeviceBuffer, ntt as _ntt
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 1 << log_n
    rng = np.random.default_rng(2024)
    x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 59) - 1)
    d = DeviceBuffer.from_numpy(c_layout_slice(x, rank, world))
    try:
        ntt_dist_dev(d, d, log_n)
        got = d.download((n // world, 4))
    finally:
        d.free()
    return bool(np.array_equal(got, _ntt(x)[m_layout_indices(n, rank, world)]))


def c_layout_slice(x, rank, world):
    """this rank's cyclic slice of a coefficient vector: x[rank + world * j]"""
    return np.ascontiguousarray(x[rank::world])


def m_layout_indices(n, rank, world):
    """global indices k of the evaluations this rank holds, in local order: local[k1 * (m / G) + t] = X[rank * (m / G) + t + m * k1]"""
    m = n // world
    chunk = m // world
    k1, t = np.divmod(np.arange(m, dtype=np.int64), chunk)
    return rank * chunk + t + m * k1


def ntt_dist_dev(d_in, d_out, log_n, inverse=False):
    """mh_ntt_dist_dev on DeviceBuffers / device pointers holding n / world elements each."""
    from .api import DeviceBuffer
    p = lambda b: b.ptr if isinstance(b, DeviceBuffer) else int(b)
    _lib.check(_lib.load().mh_ntt_dist_dev(_lib.CURVE_ID, p(d_in), p(d_out), int(log_n), 1 if inverse else 0), "mh_ntt_dist_dev")


def msm_batch_sliced_dev(bases, jobs, stride, montgomery=True, combine=True):
    """jobs: [(first_index, DeviceBuffer or pointer of the LOCAL scalars, n_local)]: scalar i multiplies base first_index + i * stride."""
    from .api import DeviceBuffer
    k = len(jobs)
    first = (C.c_size_t * k)(*[int(j[0]) for j in jobs])
    ptrs = (C.c_void_p * k)(*[(j[1].ptr if isinstance(j[1], DeviceBuffer) else int(j[1])) for j in jobs])
    ns = (C.c_size_t * k)(*[int(j[2]) for j in jobs])
    out = np.zeros((k, 3 * _lib.FQ_LIMBS), dtype=np.uint64)
    _lib.check(_lib.load().mh_msm_batch_sliced_dev(bases.handle, k, first, int(stride), ptrs, ns, 1 if montgomery else 0,
                                                   1 if combine else 0, out.ctypes.data), "mh_msm_batch_sliced_dev")
    return out
