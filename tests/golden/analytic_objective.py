"""The model-free objective of ``solver_covariance.npz`` (shared by ``make_golden.solver_covariance``, which runs the
reference's covariance helpers on it, and tests/test_solver_host.py, which runs ``HipSolve``'s)."""
import numpy as np


def solver_objective(X):
    """P = 4; vectorised over the rows of X."""
    X = np.atleast_2d(np.asarray(X, float))
    A = np.array([[4.0, 1.0, 0.5, 0.0], [1.0, 3.0, 0.2, 0.3], [0.5, 0.2, 2.0, 0.1], [0.0, 0.3, 0.1, 1.5]])
    c = np.array([5.0, 12.0, 7.0, 20.0])
    D = X - c
    return 0.5 * np.einsum("si,ij,sj->s", D, A, D) + np.sum(np.exp(0.05 * X), axis=1) + 100.0
