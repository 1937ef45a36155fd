"""Empty stand-in: the reference imports seaborn (retunegan/models/loss.py:9, audio.py:12) but never calls it
on the hot path. Used only by oracle/gen_golden.py in the build container. Test infrastructure."""
