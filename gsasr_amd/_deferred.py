"""Deferred evaluation of the reference's per-call `assert scale_modify[0] == scale_modify[1]`
(utils/gaussian_splatting.py:168-170), which for a CUDA tensor is a device-to-host synchronisation per call.  Used by
gsasr_amd.gaussian_splatting (which re-exports `deferred_asserts`)."""
import atexit

import torch


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class _DeferredAsserts:
    """The reference asserts `scale_modify[0] == scale_modify[1]` on every call (:169), which for a CUDA tensor is a
    device-to-host synchronisation per call -- sixteen per training step in the reference's per-sample loop.  Here:

    * fused path (`_StepSource`): the plan's first kernel compares the pair and sets a sticky per-device word
      (`_cabi.mismatch_flag`) when it differs.  `watch()` copies that word to pinned memory every `WATCH_EVERY` calls
      (non-blocking) and examines the copy once it has landed -- no torch kernel, copy or event per call.
    * other CUDA-tensor callers (`add()`): the comparison runs on the device, its result and the two values go to
      pinned memory without blocking, and are examined at a LATER call of the API, once the copy has landed.

    Either way: the same AssertionError (with the offending values) and no pipeline drain -- but LATE: `add()` a few calls
    late; the fused path after the 1st, 2nd, 4th, 8th ... fused call on a device and then every `WATCH_EVERY` calls (a wrong
    pair is almost always wrong from the first call on, so the early looks catch it where the reference would), and the
    image of the offending call has been rendered with `scale_modify[0]` by then.  `flush()` waits for everything
    outstanding and raises; it is also registered with `atexit`, where an exception cannot propagate: the failure is printed
    and the process exits with status 1 instead of 0 (`os._exit`), so a short script with a mismatched pair cannot end
    "successfully".  Python numbers and CPU tensors are checked on the spot, exactly as in the reference."""

    RING = 256          # pinned result slots, reused round robin (allocating pinned memory per call costs more than the check)
    WATCH_EVERY = 64

    def __init__(self):
        self.pending = []
        self.ring = None
        self.next = 0
        self.watched = {}       # device -> calls since the last look at its mismatch word
        self.seen = {}          # device -> fused calls so far (the first looks come at calls 1, 2, 4, 8, ...)

    def _slot(self):
        if self.ring is None:
            self.ring = torch.empty(self.RING, 3, dtype=torch.float32, pin_memory=True)
        if len(self.pending) >= self.RING:      # every slot in flight: wait for the oldest
            self.poll_one(wait=True)
        host = self.ring[self.next]
        self.next = (self.next + 1) % self.RING
        return host

    def add(self, pair: torch.Tensor, message: str) -> None:
        """`pair` = the two values (device tensor); fails later with `message` + the values if they differ"""
        host = self._slot()
        a, b = pair[0], pair[1]
        host.copy_(torch.stack([a, b, (a == b).to(a.dtype)]).to(torch.float32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(pair.device))
        self.pending.append((ev, host, message, None))
        self.poll()

    def watch(self, dev, force: bool = False) -> None:
        """count a fused call on `dev`; every WATCH_EVERY-th one (or `force`) fetches the device's mismatch word"""
        n = self.watched.get(dev, 0) + 1
        total = self.seen.get(dev, 0) + 1
        self.seen[dev] = total
        early = total <= self.WATCH_EVERY and (total & (total - 1)) == 0      # calls 1, 2, 4, ..., WATCH_EVERY
        if n < self.WATCH_EVERY and not force and not early:
            self.watched[dev] = n
            return
        self.watched[dev] = 0
        if _capturing():
            return
        from . import _cabi
        flag = _cabi.mismatch_flag(dev)
        host = self._slot()
        host[:2].view(torch.int32).copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.pending.append((ev, host, None, flag))
        self.poll()

    def poll_one(self, wait: bool) -> None:
        ev, host, message, flag = self.pending.pop(0)
        if wait:
            ev.synchronize()
        if flag is not None:        # the sticky word of the fused path: {1 + sample index, bits of scale_modify[0]}
            word = host[:2].view(torch.int32)
            if int(word[0]) != 0:
                sample, first = int(word[0]) - 1, float(host[1])      # (word[1] holds the float's bits)
                flag.zero_()
                # (copies of the same word taken before it was re-armed report the same pair again: drop them)
                self.pending = [e for e in self.pending if e[3] is not flag]
                raise AssertionError(f"scale_modify is not the same (sample {sample} of a fused call: scale_modify[0] = {first})")
            return
        assert bool(host[2] != 0), f"{message}-[{float(host[0])}, {float(host[1])}]"

    def poll(self, wait: bool = False) -> None:
        while self.pending and (wait or self.pending[0][0].query()):
            self.poll_one(wait)

    def flush(self) -> None:
        if torch.cuda.is_available():
            for dev in list(self.watched):
                if self.watched[dev]:
                    self.watch(dev, force=True)
        self.poll(wait=True)


deferred_asserts = _DeferredAsserts()


_EXIT_FAILED = False


def _flush_at_exit():
    global _EXIT_FAILED
    try:
        deferred_asserts.flush()
    except AssertionError as e:      # (an exception in an atexit hook is printed, not raised: say it plainly, and fail the process)
        import os
        import sys
        print(f"gsasr_amd: deferred check failed at exit: {e}", file=sys.stderr)
        if os.environ.get("GSASR_AMD_DEFERRED_EXIT", "1") != "0":    # opt-out: report only, leave the exit status alone
            _EXIT_FAILED = True      # (_exit_if_failed, which runs behind every handler registered since this import, fails the process)
    except Exception:
        pass


def _exit_if_failed():
    """The exit status can only be changed with os._exit, which skips every atexit handler still to run.  This handler is
    registered BEFORE _flush_at_exit, i.e. it runs after it and after every handler registered later than this module's import
    (the user's destroy_process_group, file closes, profilers): each of them has run exactly once by then.  (Round 5 re-ran
    atexit's whole list from inside the failing handler, which ran the later-registered ones twice: ADVICE r5.)"""
    if _EXIT_FAILED:
        import os
        import sys
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(1)


atexit.register(_exit_if_failed)
atexit.register(_flush_at_exit)
