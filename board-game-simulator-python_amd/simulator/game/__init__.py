from .protocol import ConfigLike, StateLike, ActionLike  # noqa: F401
