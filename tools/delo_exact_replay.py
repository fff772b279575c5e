"""The reference's DELO recursion on one ray's coefficient dump with the 4x4 system of every step solved (a) the way the reference
does -- cofactor inverse in double precision, radtrans_integrate.f90:685-733 -- and (b) EXACTLY, in rational arithmetic on the same double
inputs (E, F, G and every matrix entry are formed in double precision as the reference forms them; only the linear solve is exact).

tools/fuzz_cameras.py uses it on a DELO pixel that no other classifier explains: where (a) and (b) part company the reference's own
pixel is an artefact of the rounding inside its cofactor inverse, not a property of the scheme, and a device that solves the step
accurately (grt_dev_phys.hpp:delo_solve, a closed form) cannot -- and should not -- reproduce it.  r04, seed 1234 camera 300: HARM3D
at 23 GHz, a ray that leaves an emitting zone into a near-vacuum one (alpha_I 1e-97, rho_V / alpha_I = 7.7e86) and enters another
emitting zone 7 points on.  The exact solution of the first boundary step keeps V = 4.98e-15 (the component along the Faraday axis);
the cofactor inverse returns -1.4e-77.  The second boundary multiplies whatever V arrives by 4e88: 2e74 against 4e11, and the
remaining optically thick steps take 49 decades off either -- enough for the reference's number, not for the exact one."""
from fractions import Fraction as Fr
import math
import numpy as np


def opacity_matrix(K7):      # radtrans_integrate.f90:735-744
    a, p = K7[:4], K7[4:]
    M = np.zeros((4, 4))
    for i in range(4):
        M[i][i] = a[0]
    M[1][0] = a[1]; M[2][0] = a[2]; M[3][0] = a[3]; M[0][1] = a[1]; M[0][2] = a[2]; M[0][3] = a[3]
    M[2][1] = -p[2]; M[3][1] = p[1]; M[1][2] = p[2]; M[1][3] = -p[1]; M[3][2] = -p[0]; M[2][3] = p[0]
    return M


def cofactor_inverse(a):      # imatrix_4, literal operation order
    (a11, a12, a13, a14), (a21, a22, a23, a24), (a31, a32, a33, a34), (a41, a42, a43, a44) = [[float(v) for v in r] for r in a]
    b = np.zeros((4, 4))
    b[0][0] = a22 * a33 * a44 + a23 * a34 * a42 + a24 * a32 * a43 - a22 * a34 * a43 - a23 * a32 * a44 - a24 * a33 * a42
    b[0][1] = a12 * a34 * a43 + a13 * a32 * a44 + a14 * a33 * a42 - a12 * a33 * a44 - a13 * a34 * a42 - a14 * a32 * a43
    b[0][2] = a12 * a23 * a44 + a13 * a24 * a42 + a14 * a22 * a43 - a12 * a24 * a43 - a13 * a22 * a44 - a14 * a23 * a42
    b[0][3] = a12 * a24 * a33 + a13 * a22 * a34 + a14 * a23 * a32 - a12 * a23 * a34 - a13 * a24 * a32 - a14 * a22 * a33
    b[1][0] = a21 * a34 * a43 + a23 * a31 * a44 + a24 * a33 * a41 - a21 * a33 * a44 - a23 * a34 * a41 - a24 * a31 * a43
    b[1][1] = a11 * a33 * a44 + a13 * a34 * a41 + a14 * a31 * a43 - a11 * a34 * a43 - a13 * a31 * a44 - a14 * a33 * a41
    b[1][2] = a11 * a24 * a43 + a13 * a21 * a44 + a14 * a23 * a41 - a11 * a23 * a44 - a13 * a24 * a41 - a14 * a21 * a43
    b[1][3] = a11 * a23 * a34 + a13 * a24 * a31 + a14 * a21 * a33 - a11 * a24 * a33 - a13 * a21 * a34 - a14 * a23 * a31
    b[2][0] = a21 * a32 * a44 + a22 * a34 * a41 + a24 * a31 * a42 - a21 * a34 * a42 - a22 * a31 * a44 - a24 * a32 * a41
    b[2][1] = a11 * a34 * a42 + a12 * a31 * a44 + a14 * a32 * a41 - a11 * a32 * a44 - a12 * a34 * a41 - a14 * a31 * a42
    b[2][2] = a11 * a22 * a44 + a12 * a24 * a41 + a14 * a21 * a42 - a11 * a24 * a42 - a12 * a21 * a44 - a14 * a22 * a41
    b[2][3] = a11 * a24 * a32 + a12 * a21 * a34 + a14 * a22 * a31 - a11 * a22 * a34 - a12 * a24 * a31 - a14 * a21 * a32
    b[3][0] = a21 * a33 * a42 + a22 * a31 * a43 + a23 * a32 * a41 - a21 * a32 * a43 - a22 * a33 * a41 - a23 * a31 * a42
    b[3][1] = a11 * a32 * a43 + a12 * a33 * a41 + a13 * a31 * a42 - a11 * a33 * a42 - a12 * a31 * a43 - a13 * a32 * a41
    b[3][2] = a11 * a23 * a42 + a12 * a21 * a43 + a13 * a22 * a41 - a11 * a22 * a43 - a12 * a23 * a41 - a13 * a21 * a42
    b[3][3] = a11 * a22 * a33 + a12 * a23 * a31 + a13 * a21 * a32 - a11 * a23 * a32 - a12 * a21 * a33 - a13 * a22 * a31
    det = a11 * b[0][0] + a21 * b[0][1] + a31 * b[0][2] + a41 * b[0][3]
    with np.errstate(all="ignore"):
        return b / det


def exact_solve(M, y):      # Gauss-Jordan in rational arithmetic on the double entries
    A = [[Fr(float(M[r][c])) for c in range(4)] + [Fr(float(y[r]))] for r in range(4)]
    for i in range(4):
        p = max(range(i, 4), key=lambda r: abs(A[r][i]))
        if A[p][i] == 0:
            return np.full(4, np.nan)
        A[i], A[p] = A[p], A[i]
        for r in range(4):
            if r != i:
                f = A[r][i] / A[i][i]
                A[r] = [A[r][c] - f * A[i][c] for c in range(5)]
    return np.array([float(A[i][4] / A[i][i]) for i in range(4)])


def delo_step(j0, K0p, j1, K1p, dx, delta, I, exact):      # radtrans_integrate.f90:811-837 (thick) / :746-793 (thin), as delo_map restates them
    K0, K1 = opacity_matrix(K0p), opacity_matrix(K1p)
    ki, ki1 = K0p[0], K1p[0]
    eye = np.eye(4)
    with np.errstate(all="ignore"):
        if delta > 1e-2:
            E = math.exp(-delta); F = 1.0 - E; G = (1.0 - (1.0 + delta) * E) / delta
            Sp, Sp1 = j0 / ki, j1 / ki1
            Kp, Kp1 = K0 / ki - eye, K1 / ki1 - eye
            mat = eye + (F - G) * Kp
            rhs = ((F - G) * Sp + G * Sp1) + (E * eye - G * Kp1) @ I
        else:
            mat = (1.0 - delta / 2.0 + delta * delta / 6.0) * eye + (0.5 * dx - 1.0 / 6.0 * dx * dx * ki) * K0
            rhs = ((0.5 * dx * j0 - 1.0 / 6.0 * dx * dx * ki * j0) + (0.5 * j1 * dx - 1.0 / 3.0 * dx * dx * ki * j1)) + \
                  ((1.0 - 0.5 * dx * ki + 1.0 / 6.0 * dx * dx * ki * ki) * eye - (0.5 * dx - 1.0 / 3.0 * dx * dx) * K1) @ I
        if not (np.isfinite(mat).all() and np.isfinite(rhs).all()):
            return np.full(4, np.nan)
        return exact_solve(mat, rhs) if exact else cofactor_inverse(mat) @ rhs


def replay(x, j, K, tau, exact):
    """Running Stokes vector at every point, far end (index n-1) first; out[k] = I at point k."""
    n = len(x)
    out = np.zeros((n, 4))
    I = np.zeros(4)
    for k in range(n - 1, 0, -1):
        I = delo_step(np.asarray(j[k - 1], float), np.asarray(K[k - 1], float), np.asarray(j[k], float), np.asarray(K[k], float),
                      float(x[k - 1] - x[k]), float(tau[k] - tau[k - 1]), I, exact)
        out[k - 1] = I
    return out


if __name__ == "__main__":
    import sys
    d = np.load(sys.argv[1])
    for exact in (False, True):
        o = replay(d["lam"], d["j"], d["K"], d["tau"], exact)
        print("exact solve" if exact else "cofactor inverse (the reference's arithmetic)")
        for k in (45, 40, 37, 36, 30, 29, 20, 10, 0):
            if k < len(o):
                print("  point %2d  I %s" % (k, np.array2string(o[k], precision=4)), ("| reference dump %s" % np.array2string(d["I"][k], precision=4)) if "I" in d else "")
