"""Game modules of the drop-in package: `connect`, `bounce` (HIP-backed), and the structural types they satisfy."""

from .protocol import ActionLike, ConfigLike, StateLike

__all__ = ["ActionLike", "ConfigLike", "StateLike"]
