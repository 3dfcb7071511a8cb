"""TEST INFRASTRUCTURE — fake of gymnasium.utils (see ../__init__.py)."""
from . import seeding  # noqa: F401
