"""MI355X-native Robust Video Matting inference (drop-in for the reference's `model` package)."""
import os

# Kernel arguments in device memory: every launch starts with a read of its argument block, and a frame is a chain of
# ~54 dependent launches.  The HIP runtime of this stack already defaults to it on gfx950; with it forced off the HD
# frame rate drops from 2573 to 2064 frames/s (profiles/r02/README.md).  Read when HIP initialises, i.e. before the
# first GPU call of the process; a value the user set is left alone.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .model import MattingNetwork  # noqa: E402,F401

__all__ = ["MattingNetwork"]
