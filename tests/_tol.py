"""Tolerances shared by the GPU parity tests.

Covariance-like matrices are compared against each MATRIX's own scale: the device sums (p - mu)(p - mu)^T in two passes, the
reference order is a streaming Welford update — the two agree to ~1e-13 of the matrix's largest entry, but a single entry may cancel
to (almost) nothing, so an element-wise relative tolerance says nothing about it."""
import numpy as np

SIGMA_TOL = 1e-9  # |dSigma| / max|Sigma| per matrix.  Measured: 1.0e-13 on a 20 M-point campus cloud (1418 voxels of up to 100 k
# points), up to 3.3e-10 on the 100 M-point tunnel pair (2 m voxels of up to ~10^6 points: the streaming Welford update of the reference
# order loses more digits than the two-pass sum as the population grows)


def assert_sigma_close(got, exp, tol=SIGMA_TOL):
    got, exp = np.asarray(got, float), np.asarray(exp, float)
    g, e = got.reshape(got.shape[0], -1), exp.reshape(exp.shape[0], -1)
    scale = np.maximum(np.abs(e).max(axis=1, keepdims=True), 1e-300)
    err = np.abs(g - e) / scale
    assert err.max() < tol, f"covariance differs by {err.max():.3e} of its scale"


def assert_voxel_rows_close(got, exp, rtol=1e-9):
    """27-column rows of voxel_errors.txt (map_eval.cpp:292-302): bounds, mu_est, W, n_gt, n_est, Sigma_est(6), mu_gt, Sigma_gt(6)."""
    got, exp = np.asarray(got, float), np.asarray(exp, float)
    assert got.shape == exp.shape
    plain = [c for c in range(27) if not (12 <= c < 18 or 21 <= c < 27)]
    np.testing.assert_allclose(got[:, plain], exp[:, plain], rtol=rtol, atol=1e-12)
    assert np.array_equal(got[:, 10:12], exp[:, 10:12])
    assert_sigma_close(got[:, 12:18], exp[:, 12:18])
    assert_sigma_close(got[:, 21:27], exp[:, 21:27])
