"""No-op stand-in for tensorboardX.SummaryWriter (retunegan/train.py:14). Only oracle/gen_golden.py uses it."""


class SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None
