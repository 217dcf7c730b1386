"""Host logic of the kernels whose waves note their queue reservations in a finite log (rt_prep.h plan_wave_logs,
rt_hip.hip ItemPool::refill): the sizes the host picks must let the waves work off the whole queue however the
reservations fall -- also when one wave takes all it can before the others take any.  The device's reservation rule
is restated here and run against the plan on random queues."""
import ctypes as C

import numpy as np
import pytest

from tests import emu_ffi

WINDOW = 64
GUIDE = 16  # rt_hip.hip kGuide


def plan(queue_items, waves, batch_items, batch_min, log_entries, exact, quantum_windows=1):
    lib = emu_ffi.load()
    lib.emu_plan_wave_logs.restype = None
    lib.emu_plan_wave_logs.argtypes = [C.c_uint32] * 5 + [C.c_int, C.c_uint32, C.c_void_p]
    out = (C.c_uint32 * 3)()
    lib.emu_plan_wave_logs(queue_items, waves, batch_items, batch_min, log_entries, 1 if exact else 0, quantum_windows, out)
    return bool(out[0]), int(out[1]), int(out[2])


class Wave:
    """ItemPool::refill of the kernels with a log: the reservation sizes and when a wave stops."""

    def __init__(self):
        self.reserved_end = 0
        self.last_size = 0
        self.refills = 0
        self.done = False

    def reserve(self, counter, queue_items, waves, batch_items, batch_min, log_entries, quantum=WINDOW):
        size = batch_items
        if batch_min < batch_items:
            seen = self.reserved_end + waves * self.last_size
            left = queue_items - seen if seen < queue_items else 0
            want = left // (GUIDE * waves) // quantum * quantum
            size = min(max(want, batch_min), batch_items)
        self.last_size = size
        if self.refills >= log_entries:  # the log is full: sent past the end of the queue
            self.refills += 1
            self.done = True
            return counter
        base = counter
        counter += size
        self.refills += 1
        self.reserved_end = base + size
        if base >= queue_items:
            self.done = True
        return counter


def run_queue(queue_items, waves, batch_items, batch_min, log_entries, pick, quantum=WINDOW):
    pool = [Wave() for _ in range(waves)]
    counter = 0
    live = list(range(waves))
    while live:
        k = pick(live)
        before = counter
        counter = pool[k].reserve(counter, queue_items, waves, batch_items, batch_min, log_entries, quantum)
        assert before % quantum == 0 and (counter - before) % quantum == 0  # whole pixels, starting at one
        if pool[k].done:
            live.remove(k)
    return counter


@pytest.mark.parametrize("seed", range(40))
def test_the_logs_hold_every_reservation(seed):
    rng = np.random.default_rng(seed)
    waves = int(rng.choice([4, 16, 64, 256]))
    windows = int(rng.integers(1, 40 * waves))
    queue_items = windows * WINDOW - int(rng.integers(0, WINDOW))
    batch_items = int(rng.choice([1, 2, 4, 16])) * WINDOW
    batch_min = min(batch_items, int(rng.choice([1, 2])) * WINDOW)
    exact = bool(rng.integers(2))
    log_entries = int(rng.choice([2, 3, 5, 17])) if exact else int(rng.choice([65, 96, 384]))
    qw = int(rng.choice([1, 1, 3, 5]))  # windows of the smallest reservation that holds whole pixels
    quantum = qw * WINDOW
    ok, items, smallest = plan(queue_items, waves, batch_items, batch_min, log_entries, exact, qw)
    if not ok:
        # refused only when even reservations of 64 windows would not fit
        need = max(-(-windows // (waves * (log_entries - 1))), 0 if exact else -(-4 * windows // (waves * (log_entries - 64))))
        assert -(-need // qw) * qw > 64
        return
    assert items % quantum == 0 and smallest % quantum == 0 and quantum <= smallest <= items <= 64 * WINDOW
    assert items >= batch_items and smallest >= batch_min
    assert waves * (log_entries - 1) * (smallest // WINDOW) >= windows
    if not exact:
        assert 4 * windows <= waves * (log_entries - 64) * (items // WINDOW)
    # waves in random order; one wave after the other, each taking all it can; always the wave with the fullest log
    for pick in (lambda live: live[int(rng.integers(len(live)))], lambda live: live[0], lambda live: live[-1]):
        assert run_queue(queue_items, waves, items, smallest, log_entries, pick, quantum) >= queue_items


def test_the_frames_of_the_baseline():
    # 1920x1080 16x16 on 4096 waves: the usual four windows, single windows at the end
    assert plan(1920 * 1080 * 16, 4096, 256, 64, 384, False) == (True, 256, 64)
    # 3840x2160 32x32 on 3840 waves: 4 million windows, reservations of 14
    ok, items, smallest = plan(3840 * 2160 * 32, 3840, 256, 128, 384, False)
    assert ok and items == 14 * 64 and smallest == 3 * 64
    # glass_spheres 1024x768 10x10 in a stream of frames: reservations of five windows (32 whole pixels) throughout
    assert plan(1024 * 768 * 10, 4096, 256, 128, 384, False, 5) == (True, 320, 320)
    # 16384x16384 at 64 samples per row: the logs cannot hold it
    assert plan(16384 * 16384 * 8, 4096, 256, 64, 384, False)[0] is False
