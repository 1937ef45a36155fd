"""`from models import *` — the import surface retunegan/train.py:18 relies on (retunegan/models/__init__.py:1-3):
generators, discriminators, loss functions, plus the names the reference leaks through its star imports
(hp, torch, F, LRELU_SLOPE, PI, checkpoint helpers, get_param_cnt ...)."""
from .generator import *      # noqa: F401,F403
from .discrminator import *   # noqa: F401,F403
from .loss import *           # noqa: F401,F403
