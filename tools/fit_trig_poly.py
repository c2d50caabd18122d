#!/usr/bin/env python3
"""CPU: the short sin / cos polynomials of the float32 state modes (copterstep_internal.h: trig_constants, slots 16..24)
against a true minimax fit of the same length, re-derived by a Remez exchange in extended precision.

    python3 tools/fit_trig_poly.py

Prints the minimax coefficients and worst-case errors on |y| <= pi/4 for 4 / 5 sin and 5 / 6 cos coefficients, then the
shipped coefficients and the minimax ones evaluated in float64 as the kernel evaluates them (dev_math.h:
sincos_kernel<false>) over the whole interval and over |y| < 0.3, where a flying env's roll and pitch live (DESIGN
section 3: the shipped fit is the better one there)."""
import numpy as np
from numpy.polynomial import polynomial as P
L = np.longdouble
def target_sin(z):
    # S(z) = (sin(y)-y)/y^3 via series in longdouble
    s = L(0); term = L(-1)/L(6); k = 1
    out = np.zeros_like(z); t = np.full_like(z, L(-1)/L(6))
    n = 1
    acc = t.copy()
    for n in range(1, 20):
        t = t * (-z) / L((2*n+2)*(2*n+3))
        acc = acc + t
    return acc
def target_cos(z):
    # C(z) = (cos(y)-1)/z
    t = np.full_like(z, L(-1)/L(2)); acc = t.copy()
    for n in range(1, 20):
        t = t * (-z) / L((2*n+1)*(2*n+2))
        acc = acc + t
    return acc
def remez(target, weight, ncoef, zmax, iters=60):
    # minimise max |weight(z) * (p(z) - target(z))|
    k = np.arange(ncoef+1)
    x = (L(zmax)/2)*(1 - np.cos(np.pi*k/ncoef, dtype=L))
    x[0] = L(zmax)*L(1e-6)
    zz = np.linspace(L(zmax)*L(1e-7), L(zmax), 200001, dtype=L)
    for it in range(iters):
        A = np.zeros((ncoef+1, ncoef+1), dtype=L)
        for j in range(ncoef): A[:, j] = x**j
        A[:, ncoef] = ((-1)**k) / weight(x)
        b = target(x)
        # solve in longdouble by gaussian elimination
        M = np.hstack([A, b[:, None]])
        n = ncoef+1
        for i in range(n):
            p = i + np.argmax(np.abs(M[i:, i])); M[[i, p]] = M[[p, i]]
            M[i] = M[i] / M[i, i]
            for r in range(n):
                if r != i: M[r] = M[r] - M[r, i]*M[i]
        sol = M[:, -1]
        c, E = sol[:ncoef], sol[ncoef]
        err = weight(zz)*(sum(c[j]*zz**j for j in range(ncoef)) - target(zz))
        # find extrema: local maxima of |err| between sign changes
        sgn = np.sign(err); idx = np.where(sgn[1:] != sgn[:-1])[0]
        bounds = np.concatenate([[0], idx+1, [len(zz)]])
        newx = []
        for a, bb in zip(bounds[:-1], bounds[1:]):
            seg = np.abs(err[a:bb]); newx.append(zz[a+np.argmax(seg)])
        if len(newx) != ncoef+1:
            break
        newx = np.array(newx, dtype=L)
        if np.max(np.abs(newx - x)) < 1e-9*zmax: x = newx; break
        x = newx
    return c, float(np.max(np.abs(err))), float(abs(E))
zmax = (np.pi/4)**2 * 1.0
zmax = 0.7854**2
for nc in (4, 5):
    c, e, E = remez(target_sin, lambda z: z*np.sqrt(z), nc, zmax)
    print("sin", nc, e, E, [float(v).hex() for v in c])
for nc in (5, 6):
    c, e, E = remez(target_cos, lambda z: z, nc, zmax)
    print("cos", nc, e, E, [float(v).hex() for v in c])

print("---- evaluated in float64, shipped (old) vs minimax (new) coefficients: max and rms absolute error")
L=np.longdouble
rng=np.random.default_rng(0)
y=rng.uniform(-0.785398164,0.785398164,4_000_000)
z=y*y
def sinp(c):
    ps=z*c[3]+c[2]; ps=z*ps+c[1]; ps=z*ps+c[0]
    return (y*z)*ps+y
def cosp(c):
    pc=z*c[4]+c[3]; pc=z*pc+c[2]; pc=z*pc+c[1]; pc=z*pc+c[0]
    return z*pc+1.0
H=float.fromhex
old_s=[H(h) for h in ('-0x1.555555545e43fp-3','0x1.11110def9bd00p-7','-0x1.a013a80b71025p-13','0x1.6dbe28b3498d4p-19')]
new_s=[H(h) for h in ('-0x1.555555480c082p-3','0x1.111106203b4b5p-7','-0x1.a00e0e1bbf128p-13','0x1.6c8986508474fp-19')]
old_c=[H(h) for h in ('-0x1.fffffffffe699p-2','0x1.5555555150044p-5','-0x1.6c16bae67d7a6p-10','0x1.a012993437ed5p-16','-0x1.2474f436b27edp-22')]
new_c=[H(h) for h in ('-0x1.ffffffffebfe8p-2','0x1.55555546f5984p-5','-0x1.6c16b4013defcp-10','0x1.a00f12075a986p-16','-0x1.23d819969aa00p-22')]
rs=np.sin(y.astype(L)); rc=np.cos(y.astype(L))
for nm,c in (("old",old_s),("new",new_s)):
    e=np.abs(sinp(c).astype(L)-rs); print("sin",nm,float(e.max()),float(np.sqrt((e**2).mean())))
for nm,c in (("old",old_c),("new",new_c)):
    e=np.abs(cosp(c).astype(L)-rc); print("cos",nm,float(e.max()),float(np.sqrt((e**2).mean())))
# typical range |y|<0.3
m=np.abs(y)<0.3
for nm,c in (("old",old_s),("new",new_s)):
    e=np.abs(sinp(c).astype(L)-rs)[m]; print("sin<0.3",nm,float(e.max()),float(np.sqrt((e**2).mean())))
for nm,c in (("old",old_c),("new",new_c)):
    e=np.abs(cosp(c).astype(L)-rc)[m]; print("cos<0.3",nm,float(e.max()),float(np.sqrt((e**2).mean())))
