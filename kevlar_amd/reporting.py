"""Run-time reporting shared by the drivers: named stopwatches and progress lines whose frequency decays.

`Timer` and `ProgressIndicator` keep the surface the reference's drivers use (kevlar.Timer: start / stop / probe
with the unnamed watch as the total; kevlar.ProgressIndicator: message with a {counter} field, interval, breaks,
usetimer), so the log lines of a run read the same."""
from bisect import bisect_right
from time import perf_counter

import kevlar_amd


class Timer(object):
    """Stopwatches by name; the unnamed one is the total of a run."""

    def __init__(self):
        self._began = {}

    def _need(self, key):
        key = key or ''
        if key not in self._began:
            raise ValueError('No timer started for "{}"'.format(key))
        return key

    def start(self, key=None):
        key = key or ''
        if key in self._began:
            raise ValueError('Timer already started for "{}"'.format(key))
        self._began[key] = perf_counter()

    def probe(self, key=None):
        return perf_counter() - self._began[self._need(key)]

    stop = probe          # a stopped watch is never read again by the drivers: stopping is a last probe


def _milestones(first, widen_at):
    """10, 20 .. 100, 200 ..: each gap is the largest break point already reached, `first` before any."""
    due = first
    while True:
        yield due
        reached = bisect_right(widen_at, due)
        due += widen_at[reached - 1] if reached else first


class ProgressIndicator(object):
    """A line is due every `interval` items; when the count reaches one of `breaks` the interval becomes that count
    (10, 20 .. 100, 200 .. 1000 ..).  The due counts do not depend on how the caller advances, so they come from a
    generator and `update(n)` -- the batch drivers advance by thousands of reads per call -- drains the ones it
    has passed (the reference ticks once per item, kevlar/progress.py:30-42)."""

    def __init__(self, message, interval=10, breaks=(100, 1000, 10000), usetimer=False):
        self.message = message
        self.counter = 0
        self._schedule = _milestones(interval, sorted(breaks))
        self._due = next(self._schedule)
        self._clock = Timer() if usetimer else None
        if usetimer:
            self._clock.start()

    def update(self, n=1):
        self.counter += n
        while self._due < self.counter:
            line = self.message.format(counter=self._due)
            if self._clock:
                line += ' ({:.2f} seconds elapsed)'.format(self._clock.probe())
            kevlar_amd.plog(line)
            self._due = next(self._schedule)
