"""MI355X-native Slater-determinant -> MPS sweep with TeMFpy's converter API.

Sub-modules mirror the reference package (``temfpy.slater``, ``temfpy.schmidt_utils``,
``temfpy.utils``, ``temfpy.testing``); the arithmetic runs in ``libtemfpy_hip.so``.
"""
import logging as _logging

__version__ = "0.1.0"


def setup_logging(level=_logging.INFO):
    """Same helper as ``temfpy.setup_logging`` (temfpy/__init__.py:12-15)."""
    _logging.basicConfig(level=level)
