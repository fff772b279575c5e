"""tools/hermite_study.py [nup ...]: the host study behind grt_phi_kernel (r05).  The u side of the azimuth of four-real-root rays as a
two-point Hermite (Obreshkov) quadrature over I between neighbouring stored points, orders 3-5, against the one-point Carlson evaluation of
the product's own device functions compiled for the host (tests/hostsim: hs_phi_debug); six cameras, several nup.  DETAIL=1: the worst
segment of the headline camera (the reference's dropped end-point terms ON the turning point, geokerr_wrapper.f:2200).  Test tool: builds
/tmp/libhostsim_exp.so from tests/hostsim/hostsim.cpp."""
import os, sys, subprocess, ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
so = "/tmp/libhostsim_exp.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(ROOT, "grtrans_amd", "csrc"), "-o", so,
                       os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp"), os.path.join(ROOT, "grtrans_amd", "csrc", "grt_harm_loader.cpp")])
L = C.CDLL(so)
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
def G(a, l, q2, u, sg):
    c2 = a*a - q2 - l*l; c3 = 2*((a-l)**2 + q2); c4 = -a*a*q2
    U = 1 + c2*u**2 + c3*u**3 + c4*u**4
    U1 = 2*c2*u + 3*c3*u**2 + 4*c4*u**3
    U2 = 2*c2 + 6*c3*u + 12*c4*u**2
    U3 = 6*c3 + 24*c4*u
    s = np.sqrt(1 - a*a); rp = 1 + s; rm = 1 - s
    A = (l*rp + 2*(a-l))/(rp-rm); B = (l*rm + 2*(a-l))/(rm-rp)
    ip_ = 1/(1 - rp*u); im_ = 1/(1 - rm*u)
    g0 = A*ip_ + B*im_
    g1 = A*rp*ip_**2 + B*rm*im_**2
    g2 = 2*(A*rp**2*ip_**3 + B*rm**2*im_**3)
    g3 = 6*(A*rp**3*ip_**4 + B*rm**3*im_**4)
    g4 = 24*(A*rp**4*ip_**5 + B*rm**4*im_**5)
    sU = sg*np.sqrt(np.maximum(U, 0))
    G0 = g0
    G1 = g1*sU
    G2 = g2*U + g1*U1/2
    G3 = sU*(g3*U + 1.5*g2*U1 + 0.5*g1*U2)
    # G4 = d/dI G3 = d/dI[ sU * F ], F = g3 U + 1.5 g2 U1 + .5 g1 U2 ; d(sU)/dI = U1/2 ; dF/dI = sU F'
    F = g3*U + 1.5*g2*U1 + 0.5*g1*U2
    F1 = g4*U + g3*U1 + 1.5*g3*U1 + 1.5*g2*U2 + 0.5*g2*U2 + 0.5*g1*U3
    G4 = U1/2*F + U*F1
    return G0, G1, G2, G3, G4
cams = [(0.998, 0.906, (-40., 20., -20., 40.), (100, 100), 0.01, 3), (0.9375, 0.6428, (-13., 13., -13., 13.), (64, 64), 0.04, 3),
        (0.5, -0.3, (-20., 20., -20., 20.), (48, 48), 0.02, 3), (0.998, 0.1, (-25., 25., -25., 25.), (48, 48), 0.04, 3),
        (-0.9, 0.99, (-8., 8., -8., 8.), (64, 64), 0.04, 3), (0.1, 0.0, (-8., 8., -8., 8.), (64, 64), 0.04, 3)]
for nup in [int(x) for x in sys.argv[1:]] or [400, 100, 40]:
  for spin, mu0, gv, nn2, uout, stride in cams:
    nn = (nn2[0], nn2[1], nup)
    cam = (C.c_char * L.hs_sizeof_cam())()
    gva = np.array(gv, dtype=float); nna = np.array(nn, dtype=np.int32)
    L.hs_make_cam(cam, 1, dp(gva), ip(nna), C.c_double(spin), C.c_double(mu0), C.c_double(uout), C.c_double(-0.5))
    maxn = nn[2]*2 + 64
    rows = []
    for i in range(1, nn[0]*nn[1] + 1, stride):
        out = np.zeros((maxn, 8)); consts = np.zeros(16)
        n = L.hs_phi_debug(cam, i, maxn, dp(out), dp(consts))
        if n <= 1 or consts[5] == 5: continue
        a, l, q2, su, sm = consts[:5]
        I, u, mu, tpr, tpm, dphi, isfill, phiu = out[:n].T
        dI = np.diff(I)
        sg = np.where(tpr == 0, 1.0, -1.0) * np.sign(dI.mean())
        G0, G1, G2, G3, G4 = G(a, l, q2, u, sg)
        h = dI; ex = np.diff(phiu)
        s0 = G0[:-1]+G0[1:]; d1 = G1[:-1]-G1[1:]; s2 = G2[:-1]+G2[1:]; d3 = G3[:-1]-G3[1:]; s4 = G4[:-1]+G4[1:]
        T2 = h/2*s0 + h*h/12*d1
        T3 = h/2*s0 + h*h/10*d1 + h**3/120*s2
        T4 = h/2*s0 + 3*h*h/28*d1 + h**3/84*s2 + h**4/1680*d3
        T5 = h/2*s0 + h*h/9*d1 + h**3/72*s2 + h**4/1008*d3 + h**5/30240*s4
        e3 = np.abs(T3-ex); e4 = np.abs(T4-ex); e5 = np.abs(T5-ex)
        k = np.argmax(e3)
        rows.append((e3.sum(), e4.sum(), e5.sum(), e3.max(), np.abs(T3-T2).max(), np.abs(T4-T3).max(), np.abs(h).max(), i, np.abs(phiu).max(), consts[9]-consts[8] if consts[5]==7 else consts[10]-consts[9]))
    r = np.array(rows)
    q = lambda c, p: np.quantile(r[:, c], p)
    print("nup %3d a=%6g mu0=%6g rays %4d | cum err H3: med %.1e 99%% %.1e max %.1e | H4: med %.1e 99%% %.1e max %.1e | H5: med %.1e max %.1e | max|T3-T2| at worst: %.1e |T4-T3| %.1e maxh %.2f" %
          (nup, spin, mu0, len(r), q(0, .5), q(0, .99), r[:, 0].max(), q(1, .5), q(1, .99), r[:, 1].max(), q(2, .5), r[:, 2].max(), r[np.argmax(r[:, 0]), 4], r[np.argmax(r[:, 0]), 5], r[np.argmax(r[:, 0]), 6]))

if os.environ.get("DETAIL"):
    spin, mu0, gv, nn2, uout, stride = cams[0]
    nn = (nn2[0], nn2[1], 400)
    cam = (C.c_char * L.hs_sizeof_cam())()
    gva = np.array(gv, dtype=float); nna = np.array(nn, dtype=np.int32)
    L.hs_make_cam(cam, 1, dp(gva), ip(nna), C.c_double(spin), C.c_double(mu0), C.c_double(uout), C.c_double(-0.5))
    maxn = nn[2]*2 + 64
    best = None
    for i in range(1, nn[0]*nn[1] + 1, 3):
        out = np.zeros((maxn, 8)); consts = np.zeros(16)
        n = L.hs_phi_debug(cam, i, maxn, dp(out), dp(consts))
        if n <= 1 or consts[5] == 5: continue
        a, l, q2, su, sm = consts[:5]
        I, u, mu, tpr, tpm, dphi, isfill, phiu = out[:n].T
        dI = np.diff(I)
        sg = np.where(tpr == 0, 1.0, -1.0) * np.sign(dI.mean())
        G0, G1, G2, G3, G4 = G(a, l, q2, u, sg)
        h = dI; ex = np.diff(phiu)
        s0 = G0[:-1]+G0[1:]; d1 = G1[:-1]-G1[1:]; s2 = G2[:-1]+G2[1:]
        T3 = h/2*s0 + h*h/10*d1 + h**3/120*s2
        e3 = np.abs(T3-ex)
        if best is None or e3.max() > best[0]: best = (e3.max(), i, np.argmax(e3), n, consts.copy(), out[:n].copy(), T3, ex)
    e, i, k, n, consts, o, T3, ex = best
    print("worst ray", i, "segment", k, "of", n, "err", e, "consts", consts[:12])
    for j in range(max(0, k-3), min(n-1, k+4)):
        print(j+1, "I %.15g u %.15g tpr %d fill %d phiu %.15g  seg: T3 %.6e ex %.6e" % (o[j,0], o[j,1], o[j,3], o[j,6], o[j,7], T3[j], ex[j]))
