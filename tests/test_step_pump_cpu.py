"""Host logic of the streaming look-ahead (`_StepPump`, modeling_parler_tts.py) against a simulated GPU queue: the decoder graph
may run at most `max_ahead` steps past the oldest chunk boundary nobody has looked at yet (16 before the first boundary — the
time-to-first-audio case, profiles/r02_ttfa_probe.txt — a whole chunk afterwards), every step is enqueued exactly once, and the
boundaries come back in order at the streamer's `play_steps` positions."""
import pytest
import torch

from parler_tts_amd import modeling_parler_tts as M


class _SimGpu:
    """Executes one enqueued step per `tick()`; events complete when the steps enqueued before them have run."""

    def __init__(self):
        self.enqueued = 0
        self.done = 0
        self.max_lead = 0      # max (enqueued - position of the oldest unobserved boundary)
        self.calls = []

    def tick(self, n=1):
        self.done = min(self.enqueued, self.done + n)


class _Eng:
    def __init__(self, gpu):
        self.gpu = gpu

    def decode_steps(self, n):
        assert n > 0
        self.gpu.enqueued += n
        self.gpu.calls.append(n)


def _patch_events(monkeypatch, gpu, speed):
    class Ev:
        def record(self, stream):
            self.at = gpu.enqueued

        def query(self):
            gpu.tick(speed)        # the GPU makes progress while the host polls
            return gpu.done >= self.at

        def synchronize(self):
            gpu.done = max(gpu.done, self.at)

    monkeypatch.setattr(torch.cuda, "Event", Ev)


@pytest.mark.parametrize("first,chunk,remaining,speed", [(42, 43, 128, 1), (42, 43, 128, 0), (0, 10, 35, 1), (5, 16, 5, 1), (9, 10, 200, 3), (42, 43, 42, 1)])
def test_boundaries_lookahead_and_step_count(monkeypatch, first, chunk, remaining, speed):
    gpu = _SimGpu()
    _patch_events(monkeypatch, gpu, speed)
    pump = M._StepPump(_Eng(gpu), object(), first, chunk, remaining)
    expect = []
    pos, left = min(first, remaining), remaining - min(first, remaining)
    expect.append(pos)
    while left > 0:
        n = min(chunk, left)
        pos += n; left -= n
        expect.append(pos)
    seen = []
    while pump.wait_boundary():
        b = expect[len(seen)]
        assert gpu.done >= b, "boundary reported before its steps ran"
        lead = gpu.enqueued - b
        assert lead <= (16 if not seen else max(16, chunk)) + 3, (len(seen), lead)  # pieces of 4 steps: the bound may be overshot by < 4
        seen.append(b)
    assert seen == expect
    assert gpu.enqueued == remaining and all(0 < n <= 4 for n in gpu.calls)


def test_stop_ends_the_enqueueing(monkeypatch):
    gpu = _SimGpu()
    _patch_events(monkeypatch, gpu, 1)
    pump = M._StepPump(_Eng(gpu), object(), 9, 10, 1000)
    n = 0
    while pump.wait_boundary():
        n += 1
        if n == 3:
            pump.stop()
            at_stop = gpu.enqueued
    assert gpu.enqueued == at_stop < 1000       # nothing enqueued after stop(); boundaries already in flight are still reported
    assert 3 <= n <= 5
