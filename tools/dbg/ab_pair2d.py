"""dev: bench.py with the generator step's spectrogram discriminators run UNPAIRED (two half-batch passes, the behaviour before
round 5's PairConv2dFn).  usage: python tools/dbg/ab_pair2d.py <bench.py args>"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
from models import discrminator as D  # noqa: E402
_old = D._pairable
D._pairable = lambda d: _old(d) and all(c._layer.kind == 'conv' for c in list(d.convs) + [d.conv_post])
import bench  # noqa: E402
bench.main()
