"""MI355X-native Robust Video Matting inference (drop-in for the reference's `model` package)."""
from .model import MattingNetwork  # noqa: F401

__all__ = ["MattingNetwork"]
