"""Device-synchronising counterpart of the reference's Timer (timer.py:4-32): a class-level
dict of named wall-clock context managers.  The reference's version never synchronises the
device, so GPU work was only timed where a later host copy happened inside the block; here
``__enter__``/``__exit__`` synchronise the current HIP stream (``sync=False`` restores the
reference behaviour)."""
import time

import torch


class Timer(object):
    timers = {}

    def __init__(self, name, print_=False, sync=True):
        self.start = self.end = 0
        self.name = name
        self.print_ = print_
        self.sync = sync

    @classmethod
    def new(cls, name, print_=False, sync=True):
        cls.timers[name] = Timer(name, print_, sync)
        return cls.timers[name]

    def _sync(self):
        if self.sync and torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()

    def __enter__(self):
        self._sync()
        self.start = time.time()
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self._sync()
        self.end = time.time()
        if self.print_:
            print('%s: %.6fs' % (self.name, self.end - self.start))

    @classmethod
    def get(cls, name):
        return (cls.timers[name].end - cls.timers[name].start) if name in cls.timers else 0

    @classmethod
    def reset(cls):
        cls.timers = {}
