"""TEST-ONLY stand-in (see ../__init__.py)."""
from . import seeding  # noqa: F401
