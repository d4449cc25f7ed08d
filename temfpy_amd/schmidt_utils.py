"""Truncation policy with the reference's semantics (temfpy/schmidt_utils.py:18-208).

``lowest_sums`` itself runs natively (``tmf_cut_vectors`` in ``csrc/host_enum.cpp``); the
Python function here is the same entry point as ``temfpy.schmidt_utils.lowest_sums``.
"""
from __future__ import annotations

import logging
from collections.abc import Callable, Iterable
from dataclasses import dataclass
from numbers import Number

import numpy as np

logger = logging.getLogger(__name__)

_DEFAULT_SVD_MIN = 1e-6   # schmidt_utils.py:14
_DEFAULT_DEG_TOL = 1e-12  # schmidt_utils.py:15


@dataclass(frozen=True)
class StoppingCondition:
    """Same fields, defaults and checks as ``temfpy.schmidt_utils.StoppingCondition``."""

    sectors: Callable[[int], bool] | Iterable[int] | int | None = None
    chi_max: int | None = None
    svd_min: float | None = None
    degeneracy_tol: float | None = None

    def __post_init__(self):
        if self.svd_min is None:
            object.__setattr__(self, "svd_min", _DEFAULT_SVD_MIN)
        if self.degeneracy_tol is None:
            object.__setattr__(self, "degeneracy_tol", _DEFAULT_DEG_TOL)
        s = self.sectors
        if s is None:
            is_sector = lambda _: True  # noqa: E731
        elif isinstance(s, Number):
            is_sector = lambda x: x == s  # noqa: E731
        elif isinstance(s, Iterable):
            is_sector = lambda x: x in s  # noqa: E731
        elif callable(s):
            is_sector = s
        else:
            raise TypeError(f"Unexpected `sectors` parameter {s!r}")  # schmidt_utils.py:77
        object.__setattr__(self, "is_sector", is_sector)
        assert self.chi_max is None or self.chi_max > 0, \
            f"`chi_max` must be a positive integer or None, got {self.chi_max!r}"
        assert 0 < self.svd_min < 1, f"`svd_min` must be between 0 and 1, got {self.svd_min!r}"
        assert self.degeneracy_tol > 0, f"`degeneracy_tol` must be positive, got {self.degeneracy_tol!r}"
        object.__setattr__(self, "max_logval", -np.log(self.svd_min) + self.degeneracy_tol)

    def __reduce__(self):
        # picklable (worker processes of the multi-GPU path) as long as `sectors` itself is
        return (StoppingCondition, (self.sectors, self.chi_max, self.svd_min, self.degeneracy_tol))

    def __call__(self, logvals) -> bool:
        """schmidt_utils.py:99-138."""
        logvals = np.asarray(logvals)
        assert logvals.ndim == 1
        if self.chi_max is not None and len(logvals) > self.chi_max:
            return False
        if logvals[-1] - logvals[0] > self.max_logval:
            return False
        return True

    def truncate(self, logvals) -> int:
        """schmidt_utils.py:140-185."""
        logvals = np.asarray(logvals)
        assert logvals.ndim == 1
        good = np.ones(len(logvals), bool)
        if self.chi_max is not None:
            good[self.chi_max:] = False
        good &= logvals - logvals[0] < -np.log(self.svd_min)
        g2 = np.ones(len(logvals), bool)
        g2[:-1] = (logvals[1:] - logvals[:-1]) > self.degeneracy_tol
        good &= g2
        return int(np.nonzero(good)[0][-1]) + 1


def to_stopping_condition(trunc_par) -> StoppingCondition:
    """schmidt_utils.py:188-208."""
    if isinstance(trunc_par, StoppingCondition):
        return trunc_par
    if isinstance(trunc_par, dict):
        return StoppingCondition(**trunc_par)
    raise TypeError(f"Expected a dictionary or a `StoppingCondition` object, got {trunc_par!r}")


def lowest_sums(a, trunc_par: StoppingCondition, *, filled_left=None, filled_right=None):
    """Subsets of ``a`` with the lowest sums (schmidt_utils.py:211-324), natively.

    ``a`` must be of the form ``log((1-e)/e)/2``; the native routine takes ``e``.
    Returns (sums, sets) in enumeration order like the reference.
    """
    from . import _native as nat

    a = np.asarray(a, float)
    assert a.ndim == 1, f"`a` must be a 1D array, got {a.ndim!r}"
    trunc_par = to_stopping_condition(trunc_par)
    e = 1.0 / (1.0 + np.exp(2.0 * a))
    k = a.size
    if filled_left is None:
        if filled_right is None:
            fl, conv = 0, (lambda q: q)
        else:  # sector labels count particles to the right (schmidt_utils.py:262-264)
            fl, conv = 0, (lambda q: filled_right + k - q)
    else:
        fl, conv = filled_left, (lambda q: q)
    sectors = None
    if trunc_par.sectors is not None:
        sectors = [q for q in range(fl, fl + k + 1) if trunc_par.is_sector(conv(q))]
    sets, lam, q, _ = nat.cut_vectors(e, fl, trunc_par.chi_max or 0, trunc_par.svd_min, trunc_par.degeneracy_tol,
                                      sectors)
    b = np.zeros((len(sets), k), bool)
    for i in range(k):
        b[:, i] = (sets[:, i // 64] >> np.uint64(i % 64)) & np.uint64(1)
    sums = np.where(b, a, 0.0).sum(axis=1)
    order = np.argsort(sums, kind="stable")
    return sums[order], b[order]
