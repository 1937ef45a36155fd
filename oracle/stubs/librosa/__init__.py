"""Minimal stand-in for librosa==0.8.1 (requirements.txt:1), absent from this image.
Only `librosa.filters.mel` is provided: the one librosa call on the hot path (retunegan/audio.py:20,158).
Used only by oracle/gen_golden.py in the build container. Test infrastructure."""
from . import filters  # noqa: F401
