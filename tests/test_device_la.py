"""The workgroup linear algebra of the generic NFR kernel's cluster path (csrc/spg_nfr_ip.hip: team_gemm on the fp64 matrix
cores, blocked / panel Cholesky, blocked triangular inverse; csrc/spg_dev_la.hpp: tridiag_eigh) against numpy, through the
test harness kernel (`spg_debug_la`) — shapes that are not multiples of the 64-wide tiles and of the 32-wide K chunks, where
the end-to-end fixtures (n = 600, 900, 1134 ...) would not look."""
import ctypes as C
import numpy as np
import pytest

from sparsifyposegraph_amd import lib as spglib

pytestmark = pytest.mark.gpu


def _run(op, M, N, K, flags, mode, A, B, Cm):
    L = spglib.load()
    L.spg_debug_la.restype = C.c_int
    A, B, Cm = (np.ascontiguousarray(x, np.float64) for x in (A, B, Cm))
    ok = C.c_int(0)
    p = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    rc = L.spg_debug_la(op, M, N, K, flags, mode, p(A), A.shape[0], A.shape[1], p(B), B.shape[0], B.shape[1], p(Cm), Cm.shape[0], Cm.shape[1], C.byref(ok))
    assert rc == 0, rc
    return A, B, Cm, ok.value


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (65, 63, 33), (1, 7, 5), (130, 200, 97), (257, 129, 260), (600, 594, 600)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_team_gemm(M, N, K, ta, tb):
    rng = np.random.default_rng(M * 1000 + N + 7 * K + ta + 2 * tb)
    opA, opB = rng.normal(size=(M, K)), rng.normal(size=(K, N))
    A = opA.T.copy() if ta else opA
    B = opB.T.copy() if tb else opB
    C0 = rng.normal(size=(M, N + 3))            # a leading dimension larger than N
    for mode in (0, 1, 2):
        _, _, got, ok = _run(0, M, N, K, ta | (tb << 1), mode, A, B, C0.copy())
        want = C0.copy()
        prod = opA @ opB
        want[:, :N] = prod if mode == 0 else (C0[:, :N] + prod if mode == 1 else C0[:, :N] - prod)
        assert ok == 1
        assert np.array_equal(got[:, N:], C0[:, N:])                   # nothing written outside the result
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()) * K


def test_team_gemm_lower_block_triangle():
    rng = np.random.default_rng(3)
    n, K = 200, 70
    A = rng.normal(size=(n, K))
    C0 = np.full((n, n), 7.0)
    _, _, got, ok = _run(0, n, n, K, 2 | 4, 0, A, A, C0.copy())          # C = A A^T on the lower block triangle
    want = A @ A.T
    bi, bj = np.arange(n)[:, None] // 64, np.arange(n)[None, :] // 64
    assert ok == 1
    assert np.abs(got - want)[bj <= bi].max() <= 1e-11
    assert (got[bj > bi] == 7.0).all()                                  # tiles strictly above the block diagonal untouched


def _spd(n, seed, cond=1e4):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    return (Q * np.geomspace(1.0, cond, n)) @ Q.T


@pytest.mark.parametrize("op", [1, 2])
@pytest.mark.parametrize("n", [64, 65, 96, 127, 200, 333, 600])
def test_cholesky_blocked_and_panel(op, n):
    S = _spd(n, n + op)
    ld = n + 5
    A = np.zeros((n, ld))
    A[:, :n] = S
    got, _, _, ok = _run(op, n, 0, 0, 0, 0, A.copy(), np.zeros((1, 1)), np.zeros((1, 1)))
    Lw = np.linalg.cholesky(S)
    assert ok == 1
    assert np.abs(np.tril(got[:, :n]) - Lw).max() <= 1e-10 * np.abs(Lw).max()
    # not positive definite: reported, no fault
    A2 = A.copy()
    A2[n // 2, n // 2] = -1.0
    _, _, _, ok = _run(op, n, 0, 0, 0, 0, A2, np.zeros((1, 1)), np.zeros((1, 1)))
    assert ok == 0


@pytest.mark.parametrize("n", [64, 65, 130, 257, 600])
def test_triangular_inverse_blocked(n):
    Lw = np.linalg.cholesky(_spd(n, 3 * n))
    lda, ldb = n + 2, n + 7
    A = np.zeros((n, lda)); A[:, :n] = Lw + np.triu(np.full((n, n), 9.0), 1)      # the strict upper part must not be read
    B = np.full((n, ldb), 5.0)
    _, got, _, ok = _run(3, n, 0, 0, 0, 0, A, B, np.zeros((1, 1)))
    want = np.linalg.inv(Lw)
    assert ok == 1
    assert np.abs(got[:, :n] - want).max() <= 1e-9 * np.abs(want).max()
    assert (np.triu(got[:, :n], 1) == 0).all() and (got[:, n:] == 5.0).all()


@pytest.mark.parametrize("n,rank_deficit", [(2, 0), (3, 0), (64, 0), (129, 0), (135, 6), (300, 0), (517, 9)])
def test_tridiagonal_ql_eigensolver(n, rank_deficit):
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.geomspace(1e-3, 1e4, n)
    lam[:rank_deficit] = 0.0                                   # an exactly degenerate null space, as a gauge leaves it
    if n > 40:
        lam[20:24] = 3.0                                       # and a repeated eigenvalue inside the spectrum
    S = (Q * lam) @ Q.T
    S = 0.5 * (S + S.T)
    ld = n + 3
    A = np.zeros((n, ld)); A[:, :n] = S
    V = np.zeros((n, ld))
    gotA, gotV, _, ok = _run(4, n, 0, 0, 0, 0, A, V, np.zeros((1, 1)))
    assert ok == 1
    w, U = np.diag(gotA[:, :n]).copy(), gotV[:, :n]
    scale = np.abs(lam).max()
    assert np.abs(np.sort(w) - np.sort(lam)).max() <= 1e-12 * scale * n
    assert np.abs(U.T @ U - np.eye(n)).max() <= 1e-12 * n                       # orthonormal
    assert np.abs(S @ U - U * w).max() <= 1e-12 * scale * n                     # A v = lambda v, column by column
