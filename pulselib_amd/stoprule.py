"""Episode stop rule of the reference trainer without its host sync, on one GPU or on one process per GPU.

scripts/Poker/trainGPU.py:27-33 ends an episode when, at every 5th step, `terminated.float().mean() > 0.8` -- a
blocking device->host read in the middle of the loop.  `LaggedDoneCount` keeps the rule and drops the wait: each
check point ("chunk") counts the finished tables on the device, a side stream sums the count, all-reduces it over the
ranks of the job and copies it to pinned host memory, and the verdict of chunk c is taken right after chunk c + lag
has been enqueued.  The lag is FIXED (default 1), not "whatever has arrived": a run is reproducible, and every rank
decides on the same chunk from the same global count, so all ranks end every episode at the same step and issue
identical sequences of collectives.  `lag=0` is the reference's blocking check.

Who exchanges the counts between ranks:
  * `exchange="shm"` (default with the nccl backend, i.e. one process per GPU on one node): every rank writes its
    count into its cache line of a POSIX shared-memory segment and reads the others' -- host memory only, inside the
    native decide() call, no collective to launch (csrc/stoprule.hip);
  * `exchange="rccl"`: the native library's own RCCL communicator, an 8-byte all-reduce on the rule's side stream (an
    event per check point hands the counts over, which costs the steps' stream a barrier each time);
  * `exchange="host"` (gloo, i.e. the CPU tests and the one-GPU rehearsal): the local count of the due chunk is read
    from pinned memory and all-reduced with torch.distributed by `over()`;
  * `exchange="local"` (one process): nothing to exchange.
The mechanics live in the native library (pulse_stoprule_*): done through torch, the chunk boundary cost more host
time than the chunk's five steps take on the GPU."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _native


class NativeComm:
    """RCCL communicator owned by the native library (pulse_comm_*), one rank per process / GPU.  The 128-byte
    unique id travels over the already initialised torch.distributed group; creation is collective."""

    def __init__(self, device, group=None):
        import torch.distributed as dist
        self._lib = _native.lib()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            _native.check(self._lib.pulse_comm_unique_id(buf), "pulse_comm_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if dist.get_backend(group) == "nccl":
            ident = ident.to(device)
        dist.broadcast(ident, src=0, group=group)
        raw = (C.c_uint8 * 128)(*ident.cpu().tolist())
        h = C.c_void_p()
        with torch.cuda.device(device):
            _native.check(self._lib.pulse_comm_create(raw, self.rank, self.world, C.byref(h)), "pulse_comm_create")
        self.handle = h

    def all_reduce_i64_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks of a device int64 tensor, in the current stream's order."""
        assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous()
        _native.check(self._lib.pulse_comm_all_reduce_i64(self.handle, t.data_ptr(), t.data_ptr(), t.numel(),
                                                          torch.cuda.current_stream(t.device).cuda_stream), "pulse_comm_all_reduce_i64")
        return t

    def close(self):
        if getattr(self, "handle", None):
            self._lib.pulse_comm_destroy(self.handle)
            self.handle = None


class ShmExchange:
    """The native shared-memory exchange (pulse_shm_*): all_sum(index, value) over the ranks of one node.  Host code
    only; the native stop rule owns one itself, this wrapper serves the HostCounts backend (CPU tests)."""

    def __init__(self, name: str, rank: int, world: int):
        self._lib = _native.lib()
        h = C.c_void_p()
        _native.check(self._lib.pulse_shm_create(name.encode(), rank, world, C.byref(h)), "pulse_shm_create")
        self.handle = h

    def all_sum(self, index: int, value: int) -> int:
        total = C.c_int64(0)
        _native.check(self._lib.pulse_shm_all_sum(self.handle, int(index), int(value), C.byref(total)), "pulse_shm_all_sum")
        return total.value

    def close(self):
        if getattr(self, "handle", None):
            self._lib.pulse_shm_destroy(self.handle)
            self.handle = None


def _dist_world(group=None) -> int:
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class HostCounts:
    """The native rule's bookkeeping (csrc/stoprule.hip: submitted / epoch_first / fixed lag) for counts the caller
    already holds on the host -- no device involved.  `LaggedDoneCount(..., backend=HostCounts(lag))` is what the
    world-size-2 gloo tests drive on CPU; the product path uses the native handle."""

    def __init__(self, lag: int = 1):
        if not 0 <= lag < 3:
            raise ValueError("need 0 <= lag < 3")
        self.lag, self.submitted, self.epoch_first, self._counts = int(lag), 0, 0, {}

    def submit_count(self, n_done: int) -> None:
        self._counts[self.submitted] = int(n_done)
        self._counts.pop(self.submitted - 4, None)
        self.submitted += 1

    def counts(self):
        c = self.submitted - 1 - self.lag
        if c < self.epoch_first:
            return 0, 0, False
        return self._counts[c], self._counts[c], True

    def drain(self) -> None:
        self.epoch_first = self.submitted

    def close(self) -> None:
        pass


class LaggedDoneCount:
    def __init__(self, device, n_tables: int, threshold: float = 0.8, lag: int = 1, n_global: int | None = None,
                 exchange: str | None = None, group=None, comm: NativeComm | None = None, backend: HostCounts | None = None):
        self.device, self.n, self.threshold, self.lag = device, int(n_tables), float(threshold), int(lag)
        self.group = group
        self.backend = backend
        world = _dist_world(group)
        if exchange is None:
            if world == 1:
                exchange = "local"
            else:
                import torch.distributed as dist
                exchange = "shm" if dist.get_backend(group) == "nccl" else "host"
        if exchange not in ("local", "host", "rccl", "shm"):
            raise ValueError(f"exchange must be 'local', 'host', 'shm' or 'rccl', got {exchange!r}")
        if backend is not None and exchange == "rccl":
            raise ValueError("a host-side count backend cannot use the RCCL exchange")
        if exchange != "local" and world == 1 and not (exchange == "rccl" and comm is not None):
            exchange = "local"        # (an explicit communicator of one rank is honoured: the single-GPU test of the RCCL path)
        self.exchange = exchange
        self.n_global = int(n_global) if n_global is not None else (self.n if exchange == "local" else None)
        if self.n_global is None:
            import torch.distributed as dist
            t = torch.tensor([self.n], dtype=torch.int64, device=device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(t, group=group)
            self.n_global = int(t.item())
        self.decisions = 0            # verdicts taken so far (the world-size-2 tests compare this across ranks)
        self.exchanges = 0            # host-side all-reduces issued by over()
        self._own_comm = None
        self.comm = None
        self.handle = None
        self._shm = None
        if backend is not None:
            if backend.lag != self.lag:
                raise ValueError("backend.lag differs from lag")
            if exchange == "shm":
                import torch.distributed as dist
                name = self._shared_name(device, group)
                self._shm = ShmExchange(name, dist.get_rank(group), world)
                dist.barrier(group)
                if dist.get_rank(group) == 0:
                    try:
                        os.unlink("/dev/shm" + name)
                    except OSError:
                        pass
            return
        self._lib = _native.lib()
        if exchange == "rccl" and comm is None:
            comm = self._own_comm = NativeComm(device, group)
        self.comm = comm if exchange == "rccl" else None
        rank, shm_name = 0, None
        if exchange == "shm":
            import torch.distributed as dist
            rank = dist.get_rank(group)
            shm_name = self._shared_name(device, group)
        h = C.c_void_p()
        with torch.cuda.device(device):
            _native.check(self._lib.pulse_stoprule_create(self.n, self.n_global, self.threshold, self.lag,
                                                          self.comm.handle if self.comm is not None else None,
                                                          shm_name.encode() if shm_name else None, rank, world if exchange == "shm" else 1,
                                                          C.byref(h)), "pulse_stoprule_create")
        self.handle = h
        if exchange == "shm":              # every rank has the segment mapped: its name can go
            import torch.distributed as dist
            dist.barrier(group)
            if rank == 0:
                try:
                    os.unlink("/dev/shm" + shm_name)
                except OSError:
                    pass

    _shm_counter = 0

    @classmethod
    def _shared_name(cls, device, group):
        """A segment name all ranks agree on: rank 0 picks it (its pid + a counter + a time stamp), removes whatever a
        crashed job may have left under that name (a stale segment would carry old sequence numbers), and only then
        broadcasts the 24 bytes -- no rank can open the name before it has been cleared."""
        import time
        import torch.distributed as dist
        cls._shm_counter += 1
        ident = torch.tensor([os.getpid(), cls._shm_counter, time.time_ns() & ((1 << 40) - 1)], dtype=torch.int64)
        if dist.get_rank(group) == 0:
            try:
                os.unlink("/dev/shm" + cls._name_of(*ident.tolist()))
            except OSError:
                pass
        if dist.get_backend(group) == "nccl":
            ident = ident.to(torch.device(device))
        dist.broadcast(ident, src=0, group=group)
        return cls._name_of(*(int(x) for x in ident.cpu().tolist()))

    @staticmethod
    def _name_of(pid, k, stamp):
        return f"/pulse_stoprule_{pid}_{k}_{stamp:x}"

    def submit(self, flags: torch.Tensor) -> None:
        """One check point: count the set flags of `flags` (bool/uint8[n]) in stream order.
        (PokerGPU.rollout(..., stop_rule=self) does this inside the native call that enqueues the steps.)"""
        if self.backend is not None:
            self.backend.submit_count(int(flags.sum()))
            return
        _native.check(self._lib.pulse_stoprule_submit(self.handle, flags.data_ptr(), flags.numel(),
                                                      torch.cuda.current_stream(self.device).cuda_stream), "pulse_stoprule_submit")

    def counts(self):
        """(local, global, have) of the chunk whose verdict is due: the one submitted `lag` chunks before the newest."""
        if self.backend is not None:
            return self.backend.counts()
        loc, glob, have = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        _native.check(self._lib.pulse_stoprule_counts(self.handle, C.byref(loc), C.byref(glob), C.byref(have)), "pulse_stoprule_counts")
        return loc.value, glob.value, bool(have.value)

    def over(self) -> bool:
        """Verdict of the due chunk (False while no chunk of this episode is due).  Identical on every rank."""
        if self.exchange == "host":
            import torch.distributed as dist
            loc, _, have = self.counts()
            if not have:                      # the same on every rank: chunk indices advance in lock step
                return False
            t = torch.tensor([loc], dtype=torch.int64)
            dist.all_reduce(t, group=self.group)
            self.decisions += 1
            self.exchanges += 1
            return float(t.item()) > self.threshold * self.n_global
        if self.backend is not None:
            loc, glob, have = self.backend.counts()
            self.decisions += 1
            if have and self._shm is not None:
                glob = self._shm.all_sum(self.backend.submitted - 1 - self.lag, loc)
                self.exchanges += 1
            return have and glob > self.threshold * self.n_global
        flag = C.c_int32(0)
        _native.check(self._lib.pulse_stoprule_decide(self.handle, C.byref(flag)), "pulse_stoprule_decide")
        self.decisions += 1
        return bool(flag.value)

    @property
    def native_mode(self) -> str | None:
        """What the native handle does with a check point's count: 'local', 'rccl' (side stream) or 'shm'."""
        if self.handle is None:
            return None
        return ("local", "rccl", "shm")[self._lib.pulse_stoprule_mode(self.handle)]

    @property
    def side_stream_check_points(self) -> int:
        return int(self._lib.pulse_stoprule_side_launches(self.handle)) if self.handle is not None else 0

    OPT_VERDICT_WAIT_TICKS, OPT_ALLOW_SHARED_DEVICE_PAIRS, OPT_DEBUG_LATE_VERDICTS = 0, 1, 2     # pulse_env.h: PULSE_STOPRULE_OPT_*

    def set_option(self, option: int, value: int) -> None:
        """Options of the native handle (pulse_stoprule_set_option): how long a paired launch waits for its verdict, whether
        ranks sharing a device may pair, the late-verdict test hook."""
        if self.handle is None:
            raise RuntimeError("set_option: no native handle (host-side count backend)")
        _native.check(self._lib.pulse_stoprule_set_option(self.handle, int(option), int(value)), "pulse_stoprule_set_option")

    def stats(self) -> dict:
        """{'paired_launches': issued so far, 'verdict_timeouts': launches that gave up waiting for this host (each made
        the handle fall back to one check interval per launch), 'pairs': whether the handle would still pair,
        'side_stream_check_points'} -- bench.py reports the first three."""
        if self.handle is None:
            return {"paired_launches": 0, "verdict_timeouts": 0, "pairs": False, "side_stream_check_points": 0}
        out = (C.c_int64 * 4)()
        _native.check(self._lib.pulse_stoprule_stats(self.handle, out), "pulse_stoprule_stats")
        return {"paired_launches": int(out[0]), "verdict_timeouts": int(out[1]), "pairs": bool(out[2]), "side_stream_check_points": int(out[3])}

    def publish(self) -> None:
        """Publish the newest check point's count now (pulse_stoprule_publish) instead of with the next launch that carries
        the rule -- for the trainer, whose next such launch is five steps away."""
        if self.handle is not None:
            _native.check(self._lib.pulse_stoprule_publish(self.handle), "pulse_stoprule_publish")

    def drain(self) -> None:
        """Episode boundary: the chunks submitted so far decide nothing any more."""
        if self.backend is not None:
            self.backend.drain()
            return
        _native.check(self._lib.pulse_stoprule_drain(self.handle), "pulse_stoprule_drain")

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.pulse_stoprule_destroy(self.handle)
            self.handle = None
        if self._own_comm is not None:
            self._own_comm.close()
            self._own_comm = None
        if getattr(self, "_shm", None) is not None:
            self._shm.close()
            self._shm = None

    # no __del__: at interpreter shutdown the HIP runtime may already be gone, and an unclosed handle only leaks
    # < 1 MB; long-lived callers call close()


class RolloutTimer:
    """HIP-event pairs around native roll-out calls on their launch stream (pulse_timer_*)."""

    def __init__(self):
        self._lib = _native.lib()
        h = C.c_void_p()
        _native.check(self._lib.pulse_timer_create(C.byref(h)), "pulse_timer_create")
        self.handle = h

    def collect(self):
        """(sum of milliseconds, launches, steps) of the calls bracketed since the last collect; sync the stream first."""
        ms, n_l, n_s = C.c_float(0), C.c_int32(0), C.c_int64(0)
        _native.check(self._lib.pulse_timer_collect(self.handle, C.byref(ms), C.byref(n_l), C.byref(n_s)), "pulse_timer_collect")
        return ms.value, n_l.value, n_s.value

    def close(self):
        if getattr(self, "handle", None):
            self._lib.pulse_timer_destroy(self.handle)
            self.handle = None
