"""A second, independent restatement of Nister's five-point solver in numpy (test helper, not collected): null space by
numpy's SVD, polynomials as exponent dictionaries, elimination by numpy.linalg.solve, roots by numpy.roots (companion
matrix eigenvalues) -- none of the oracle's choices (one-sided Jacobi, coefficient tables, Gauss-Jordan, bisection between
critical points).  tests/test_oracle_five_point.py holds oracle/five_point.c against it."""
import numpy as np
# monomials of degree <= 3 in (x, y, z) as exponent triples; Nister's column order
ORDER = [(3,0,0),(0,3,0),(2,1,0),(1,2,0),(2,0,1),(2,0,0),(0,2,1),(0,2,0),(1,1,1),(1,1,0),
         (1,0,2),(1,0,1),(1,0,0),(0,1,2),(0,1,1),(0,1,0),(0,0,3),(0,0,2),(0,0,1),(0,0,0)]
def pmul(a, b):
    """polynomials as dict exponent->coef"""
    o = {}
    for ea, ca in a.items():
        for eb, cb in b.items():
            e = (ea[0]+eb[0], ea[1]+eb[1], ea[2]+eb[2])
            o[e] = o.get(e, 0.0) + ca*cb
    return o
def padd(a, b, s=1.0):
    o = dict(a)
    for e, c in b.items(): o[e] = o.get(e, 0.0) + s*c
    return o
def five_point_np(x0, y0, x1, y1):
    q0 = np.stack([x0, y0, np.ones(5)], 1); q1 = np.stack([x1, y1, np.ones(5)], 1)
    A = np.stack([np.kron(q1[i], q0[i]) for i in range(5)])
    _, _, Vt = np.linalg.svd(A)
    X, Y, Z, W = [Vt[k].reshape(3, 3) for k in (5, 6, 7, 8)]
    E = [[{(1,0,0): X[r,c], (0,1,0): Y[r,c], (0,0,1): Z[r,c], (0,0,0): W[r,c]} for c in range(3)] for r in range(3)]
    def mm(P, Q):   # 3x3 poly matrices
        return [[padd(padd(pmul(P[r][0], Q[0][c]), pmul(P[r][1], Q[1][c])), pmul(P[r][2], Q[2][c])) for c in range(3)] for r in range(3)]
    Et = [[E[c][r] for c in range(3)] for r in range(3)]
    EEt = mm(E, Et)
    tr = padd(padd(EEt[0][0], EEt[1][1]), EEt[2][2])
    EEtE = mm(EEt, E)
    cons = []
    det = padd(padd(pmul(E[0][0], padd(pmul(E[1][1], E[2][2]), pmul(E[1][2], E[2][1]), -1.0)),
                    pmul(E[0][1], padd(pmul(E[1][0], E[2][2]), pmul(E[1][2], E[2][0]), -1.0)), -1.0),
               pmul(E[0][2], padd(pmul(E[1][0], E[2][1]), pmul(E[1][1], E[2][0]), -1.0)))
    cons.append(det)
    for r in range(3):
        for c in range(3):
            cons.append(padd({e: 2*v for e, v in EEtE[r][c].items()}, pmul(tr, E[r][c]), -1.0))
    M = np.array([[p.get(e, 0.0) for e in ORDER] for p in cons])
    R = np.linalg.solve(M[:, :10], M[:, 10:])          # rows a..j after elimination
    def row_polys(e, f):
        kx = np.array([-R[f,0], R[e,0]-R[f,1], R[e,1]-R[f,2], R[e,2]])
        ky = np.array([-R[f,3], R[e,3]-R[f,4], R[e,4]-R[f,5], R[e,5]])
        k1 = np.array([-R[f,6], R[e,6]-R[f,7], R[e,7]-R[f,8], R[e,8]-R[f,9], R[e,9]])
        return kx, ky, k1
    kx, ky, k1 = row_polys(4, 5); lx, ly, l1 = row_polys(6, 7); mx, my, m1 = row_polys(8, 9)
    P = np.polymul; S = np.polysub
    detB = np.polyadd(S(P(kx, S(P(ly, m1), P(l1, my))), P(ky, S(P(lx, m1), P(l1, mx)))), P(k1, S(P(lx, my), P(ly, mx))))
    out = []
    for z in np.roots(detB):
        if abs(z.imag) > 1e-9 * max(1, abs(z.real)): continue
        z = z.real
        v = lambda p: np.polyval(p, z)
        den = v(kx)*v(ly) - v(ky)*v(lx)
        x = (v(ky)*v(l1) - v(k1)*v(ly)) / den
        y = (v(k1)*v(lx) - v(kx)*v(l1)) / den
        Ez = x*X + y*Y + z*Z + W
        out.append(Ez / np.linalg.norm(Ez) * np.sqrt(2))
    return out
