"""dev: bench.py with the spectrogram discriminators in the reference's layout ([B, C, F, frames]; the product runs them along
the frequency axis).  usage: python tools/dbg/ab_layout.py <bench.py args>"""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'transtacos-retunegan_amd'))
sys.path.insert(0, REPO)
from rtg import ops  # noqa: E402
from models import discrminator  # noqa: E402
ops.SPEC_FREQ_MAJOR = False
discrminator.MTD_ALONG_FREQ = False
import bench  # noqa: E402
bench.main()
