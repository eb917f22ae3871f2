"""Named wall-clock sections (the reference's kevlar/timer.py:13-39)."""
import time


class Timer(object):
    def __init__(self):
        self._started = {}
        self._stopped = {}

    @staticmethod
    def _key(key):
        return '' if key is None else key

    def start(self, key=None):
        key = self._key(key)
        if key in self._started:
            raise ValueError('Timer already started for "' + key + '"')
        self._started[key] = time.time()

    def stop(self, key=None):
        key = self._key(key)
        if key not in self._started:
            raise ValueError('No timer started for "' + key + '"')
        self._stopped[key] = time.time()
        return self._stopped[key] - self._started[key]

    def probe(self, key=None):
        key = self._key(key)
        if key not in self._started:
            raise ValueError('No timer started for "' + key + '"')
        return time.time() - self._started[key]
