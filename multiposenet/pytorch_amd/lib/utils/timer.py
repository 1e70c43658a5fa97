"""tic/toc timer with a running average — the reference's ``lib/utils/timer.py`` contract (``duration`` is the average
of the intervals since the last ``clear()``)."""
import time


class Timer(object):
    def __init__(self):
        self.clear()

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.average_time = self.total_time / self.calls
        self.duration = self.average_time if average else self.diff
        return self.duration

    def clear(self):
        self.total_time = self.start_time = self.diff = self.average_time = self.duration = 0.
        self.calls = 0
