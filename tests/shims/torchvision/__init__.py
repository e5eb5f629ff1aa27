"""Minimal stand-in for ``torchvision`` (absent from this image) -- TEST FIXTURE, not product code.  The reference
needs ``torchvision.utils.save_image`` (render_hierarchy.py:24,104) and ``from torchvision import models`` to succeed
at import time (lpipsPyTorch/modules/networks.py:7; the networks are only instantiated under ``--eval``, which also
needs downloaded weights -- unavailable offline)."""
from . import models, utils  # noqa: F401

__version__ = "0.0-shim"
