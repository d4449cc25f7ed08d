"""Run-time self-checks with the reference's global switch (temfpy/testing.py:15-128)."""
from typing import Literal
import warnings

import numpy as np

_DIAG_TOL = 1e-8

TEST_ACTION: Literal["raise", "warn", "pass"] = "warn"


class ComparisonWarning(Warning):
    """Generic warning class for failed equality testing or comparison."""


def _report(fn, err_msg, *args, **kw):
    if TEST_ACTION == "raise":
        fn(*args, err_msg=err_msg, **kw)
    elif TEST_ACTION == "warn":
        try:
            fn(*args, **kw)
        except AssertionError as err:
            warnings.warn("\n" + err_msg + str(err), category=ComparisonWarning)
    elif TEST_ACTION != "pass":
        raise ValueError(f"Invalid value {TEST_ACTION!r} of `temfpy_amd.testing.TEST_ACTION`,\n"
                         "must be one of 'raise', 'warn', 'pass'.")


def _all_close(a, d, rtol, atol):
    """np.all(|a - d| <= atol + rtol |d|) for arrays of one shape, evaluated in blocks of rows that stay in the cache (the
    one-piece expression moves five 16 MB temporaries through memory for a 1024 x 1024 complex matrix: 6 ms per check, two
    checks and a Hermitisation per Pfaffian conversion).  For rtol = 0 the modulus of a complex difference is only formed where
    max(|re|, |im|) does not decide the comparison already."""
    if a.ndim != 2 or a.size < (1 << 16):
        return bool(np.all(np.abs(a - d) <= atol + rtol * np.abs(d)))
    step = max(1, (1 << 16) // a.shape[1])
    for i in range(0, a.shape[0], step):
        x = a[i: i + step] - d[i: i + step]
        if rtol == 0:
            if np.iscomplexobj(x):
                m = max(np.abs(x.real).max(), np.abs(x.imag).max())
                if m <= atol * 0.7071067811865475:
                    continue
                if m > atol:
                    return False
            if not np.all(np.abs(x) <= atol):
                return False
        elif not np.all(np.abs(x) <= atol + rtol * np.abs(d[i: i + step])):
            return False
    return True


def assert_allclose(actual, desired, rtol=1e-7, atol=0.0, equal_nan=True, err_msg="", verbose=False):
    if TEST_ACTION == "pass":
        return
    try:  # fast path: numpy.testing spends ~30 ms per 1024 x 1024 complex comparison building its report
        with np.errstate(invalid="ignore"):
            a, d = np.asarray(actual), np.asarray(desired)
            if a.shape == d.shape and _all_close(a, d, rtol, atol):
                return
    except (TypeError, ValueError):
        pass
    _report(np.testing.assert_allclose, err_msg, actual, desired, rtol=rtol, atol=atol, equal_nan=equal_nan,
            verbose=verbose)


def assert_array_less(x, y, err_msg="", verbose=False):
    _report(np.testing.assert_array_less, err_msg, x, y, verbose=verbose)


def report_schmidt_checks(deviations, diag_tol=_DIAG_TOL):
    """Outcome of ``check_schmidt_decomposition`` (testing.py:131-177) from deviations that were
    evaluated on the device: ``deviations`` maps the reference's error message to the largest
    absolute difference between a block of the correlation matrix and its reconstruction.
    Raises / warns / ignores according to :data:`TEST_ACTION`, like the reference."""
    if TEST_ACTION == "pass":
        return
    if TEST_ACTION not in ("raise", "warn"):
        raise ValueError(f"Invalid value {TEST_ACTION!r} of `temfpy_amd.testing.TEST_ACTION`,\n"
                         "must be one of 'raise', 'warn', 'pass'.")
    for err_msg, dev in deviations.items():
        if not dev <= diag_tol:
            text = (f"\n{err_msg}\nNot equal to tolerance rtol=0, atol={diag_tol:g}\n"
                    f"Max absolute difference among violations: {dev:.8g}")
            if TEST_ACTION == "raise":
                raise AssertionError(text)
            warnings.warn(text, category=ComparisonWarning)
