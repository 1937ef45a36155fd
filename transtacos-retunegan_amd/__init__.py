"""transtacos-retunegan_amd — MI355X-native RetuneGAN train-step hot path.

The directory is laid out like the reference's `retunegan/` source dir and is meant to be put on sys.path the same
way (`import hparam as hp; from models import *; from train import Trainer`): importing this package (the name has a
hyphen, so through importlib.import_module) does exactly that.

  hparam.py, utils.py, audio.py, models/, train.py   host-side mirror of the reference interface for the path
  rtg/                                               ctypes binding, weight bank, autograd wrappers
  csrc/                                              HIP kernels + the C ABI of include/rtg.h  -> librtg.so
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
