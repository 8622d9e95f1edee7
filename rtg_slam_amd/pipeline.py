"""Tracker || mapper hand-off on one GPU: the two-stage pipeline of SLAM/multiprocess/system.py:12-87 (there: two
spawned processes exchanging frames through queues, each owning a Renderer) as TWO HIP STREAMS of one process.

Tracking frame t+1 and optimising the map on frame t are independent within a step; the tracker's kernels (pyramids,
15 Gauss-Newton iterations) are short and latency-bound, the mapper's are long - run on separate streams they overlap
on the chip.  The tracker's kernels are ENQUEUED by a helper thread (the C entry points release the GIL), so neither the
GPU nor the host serialises the two stages; results cross over through stream events, not host synchronisation:

    pipe = TrackMapPipeline(device)
    pipe.track(lambda: tracker.predict_pose(frame))     # enqueued on the tracker stream, ordered after the main stream
    map_step()                                          # main stream, concurrently
    pose, ok = pipe.result()                            # joins the helper thread; the main stream now waits for the tracker's

bench.py's frame() is exactly this.  Exceptions raised by the tracker function surface in `result()`."""
from __future__ import annotations

import queue
import threading
from typing import Any, Callable, Optional

import torch


class TrackMapPipeline:
    def __init__(self, device: torch.device, tracker_stream: Optional[torch.cuda.Stream] = None, tracker_priority: int = -1,
                 reserve_cus: int = 0):
        self.device = torch.device(device)
        # reserve_cus > 0: `mapper_stream` is a stream whose CU mask leaves that many compute units free - run the mapper
        # on it (with torch.cuda.stream(pipe.mapper_stream): ...) and the tracker's short dependent kernels always find
        # wave slots instead of waiting for the mapper's workgroups to retire (include/rtgs_raster.h)
        self.mapper_stream: Optional[torch.cuda.Stream] = None
        self._masked = None
        if reserve_cus > 0:
            from . import _lib
            with torch.cuda.device(self.device):
                self._masked = _lib.load().rtgs_stream_create_reserving(int(reserve_cus))
            if not self._masked:
                raise RuntimeError(f"rtgs_stream_create_reserving({reserve_cus}) failed")
            self.mapper_stream = torch.cuda.ExternalStream(self._masked, device=self.device)
        # The tracker's kernels are short and sit on the frame's critical path (16 dependent launches); on a high-priority
        # stream they are dispatched ahead of the mapper's queued workgroups instead of waiting behind them.
        self.tracker_stream = (tracker_stream if tracker_stream is not None
                               else torch.cuda.Stream(device=self.device, priority=tracker_priority))
        # (Round 6 measured whether the rare 8.5-ms host frame of bench.py's unit loop is the GIL hand-over between this thread
        # pair: RTGS_GIL_SWITCH_US = 50 against the interpreter's 5 000 made no difference - tools/hiccup_ab.sh - so the
        # interpreter's switch interval is left alone unless the variable is set.)
        import os
        import sys
        us = float(os.environ.get("RTGS_GIL_SWITCH_US", "0"))
        if us > 0:
            sys.setswitchinterval(us * 1e-6)
        self._req: "queue.SimpleQueue[Optional[Callable[[], Any]]]" = queue.SimpleQueue()
        self._done: "queue.SimpleQueue[Any]" = queue.SimpleQueue()
        self._pending = 0
        self._thread = threading.Thread(target=self._worker, daemon=True, name="rtgs-tracker-enqueue")
        self._thread.start()

    def _worker(self):
        torch.cuda.set_device(self.device)
        while True:
            fn = self._req.get()
            if fn is None:
                return
            try:
                with torch.cuda.stream(self.tracker_stream):
                    self._done.put((True, fn()))
            except BaseException as e:          # surfaces in result()
                self._done.put((False, e))

    def track(self, fn: Callable[[], Any]) -> None:
        """Run `fn` (the tracker stage of one frame) on the tracker stream, ordered after everything the calling
        thread has enqueued on ITS current stream so far (the frame's inputs, the last map update)."""
        self.tracker_stream.wait_stream(torch.cuda.current_stream(self.device))
        self._pending += 1
        self._req.put(fn)

    def result(self) -> Any:
        """The oldest outstanding tracker result; afterwards work enqueued on the caller's current stream is ordered
        after the tracker's kernels."""
        if self._pending == 0:
            raise RuntimeError("TrackMapPipeline.result(): no tracker stage outstanding")
        ok, out = self._done.get()
        self._pending -= 1
        torch.cuda.current_stream(self.device).wait_stream(self.tracker_stream)
        if not ok:
            raise out
        return out

    def close(self) -> None:
        if self._thread.is_alive():
            self._req.put(None)
            self._thread.join(timeout=5.0)
        if self._masked:
            from . import _lib
            torch.cuda.synchronize(self.device)
            if torch.cuda.current_stream(self.device) == self.mapper_stream:
                torch.cuda.set_stream(torch.cuda.default_stream(self.device))
            _lib.load().rtgs_stream_destroy(self._masked)
            self._masked, self.mapper_stream = None, None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
