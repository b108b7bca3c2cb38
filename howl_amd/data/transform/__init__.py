from .operator import ZmuvTransform  # noqa: F401  (re-exported like howl/data/transform/__init__.py:1)
