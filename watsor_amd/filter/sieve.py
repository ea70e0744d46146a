"""`DetectionSieve` fast path (`watsor/filter/sieve.py:8-56`) for cameras whose filters run on the GPU.

The reference's sieve thread clones the frame's 100 `Detection` rows, calls every filter on them in Python and
copies the survivors back (sieve.py:21-33,38-56).  `hip_detection_sieve()` returns a subclass of the reference's
own `DetectionSieve` (so the thread / queue / latch protocol of `WorkPassthroughPublish` is reused unchanged) whose
`_incoming_frame` hands `frame.header.detections` to `HipTrackFilter.sieve()`: one native call per frame, in place in
shared memory.  Filters that are not pre-filtered trackers still go through the reference's own code path.

Needs an installed Watsor (the class is derived at call time); nothing else in this package does.
"""
from __future__ import annotations

from .track import HipTrackFilter


def hip_detection_sieve():
    from watsor.filter.sieve import DetectionSieve                # the reference, unmodified

    class HipDetectionSieve(DetectionSieve):
        def _incoming_frame(self, frame, stop_event, filters, decoder_rate_limiter, fps, *args, **kwargs):
            if len(filters) == 1 and isinstance(filters[0], HipTrackFilter):
                try:
                    suspicious_activity = filters[0].sieve(frame.header.detections)
                except ValueError:                                # tracker carries Python filters: reference path
                    return super()._incoming_frame(frame, stop_event, filters, decoder_rate_limiter, fps,
                                                   *args, **kwargs)
                if suspicious_activity:                           # sieve.py:29-31
                    if decoder_rate_limiter.unlimited():
                        self._logger.debug("FPS is unlimited due to an object detected")
                fps(value=True)                                   # sieve.py:33
                return
            return super()._incoming_frame(frame, stop_event, filters, decoder_rate_limiter, fps, *args, **kwargs)

    return HipDetectionSieve
