#!/usr/bin/env python3
"""patch_proto.py -- float64 numpy model of the patch-resident IBP iteration (csrc/srx_patch.hpp).

Development tool: states the algorithm of k_ibp_patch in "natural" coordinates and checks it against the CPU oracle
(oracle/sr_oracle.c, the restatement of mono_cal_target/run_sr.py:190-209).  Three layers, each checked against the one before:

  dense_iteration    the mosaic formulation (srx_mosaic.hpp) with dense numpy operators on padded planes
  Chain1D            the per-axis operator chains evaluated block by block (64-sample blocks, local recursions + carry
                     fix-ups, closed-form pad boundaries), exactly as the kernel's waves evaluate them
  patch_iteration    dense G step + Chain1D on both axes + wrapped extra row, i.e. the kernel's data flow

Natural coordinates (per axis): HR index i in [0, n); rho = Y index (Y[rho] is a 4-tap combination of spline coefficients
rho-2 .. rho+1 of the blurred image, or b[rho] itself when delta = 0); g = G index = p' - 13 (G[g] collects the LR samples /
residuals whose zero-inserted position is g + n_k).  Padded indices of srx_mosaic.hpp: P = rho + E, p' = g + 13.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "enph459-super-resolution_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import sr_oracle as O  # noqa: E402
from sr_mi355x import synth  # noqa: E402

Z = np.sqrt(3.0) - 2.0
NPAD = 12


def bspline3(t):
    y, z = t, 1.0 - t
    w1 = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
    w2 = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0
    w0 = z * z * z / 6.0
    return np.array([w0, w1, w2, 1.0 - w0 - w1 - w2])


class AxisPlan:
    def __init__(self, shifts, axis, f):
        d = np.array([s[axis] * f for s in shifts], dtype=np.float64)
        n = np.floor(d)
        fr = d - n
        n[fr > 1 - 1e-12] += 1
        fr[fr > 1 - 1e-12] = 0
        fr[fr < 1e-12] = 0
        assert np.all(np.abs(fr - fr[0]) < 1e-12), "no common fraction"
        self.delta = float(fr[0])
        self.zero = self.delta == 0.0
        self.n = n.astype(int)
        self.E = 11 if self.zero else 10
        self.D = 13 - self.E
        self.PB = 13 - int(self.n.min())
        self.RS = max(13 - int(self.n.max()), 0)
        self.wf = bspline3(0.0 if self.zero else 1.0 - self.delta)
        self.wb = bspline3(self.delta)


def prefilter_reflect(a, axis):
    """scipy.ndimage.spline_filter1d(order=3, mode='reflect') along `axis` (exact boundary sums, float64)."""
    a = np.moveaxis(np.array(a, dtype=np.float64), axis, 0).copy()
    n = a.shape[0]
    a *= 6.0
    zi = Z ** np.arange(n)
    s = np.tensordot(zi + Z ** (2 * n - 1 - np.arange(n)), a, axes=(0, 0))
    a[0] = a[0] + Z * s / (1.0 - Z ** (2 * n))
    for i in range(1, n):
        a[i] += Z * a[i - 1]
    a[n - 1] *= Z / (Z - 1.0)
    for i in range(n - 2, -1, -1):
        a[i] = Z * (a[i + 1] - a[i])
    return np.moveaxis(a, 0, axis)


def corr7_1d(x, k, axis):
    """zero-padded 7-tap correlation out[i] = sum_u k[u] x[i-3+u]."""
    x = np.moveaxis(x, axis, 0)
    n = x.shape[0]
    p = np.concatenate([np.zeros((3,) + x.shape[1:]), x, np.zeros((3,) + x.shape[1:])])
    out = sum(k[u] * p[u:u + n] for u in range(7))
    return np.moveaxis(out, 0, axis)


def sep_kernel(psf):
    """Kernel7 of srx_fused.hpp for a rank-1 PSF: correlation weights cy, cx (k[u, v] = cy[u] cx[v]) and the flipped pair."""
    c = psf[::-1, ::-1]  # convolution -> correlation
    um, vm = np.unravel_index(np.argmax(np.abs(c)), c.shape)
    cy, cx = c[:, vm].copy(), c[um, :] / c[um, vm]
    assert np.max(np.abs(np.outer(cy, cx) - c)) < 1e-15
    return cy, cx


# --------------------------------------------------------------------------------------------
# layer 1: dense mosaic formulation on padded planes
# --------------------------------------------------------------------------------------------
def build_tables(lr, py, px, f, H, W):
    """M, C, per-pixel lists of (rho_y, rho_x) in NATURAL coordinates, Mu / Cu, Vtot.  G plane indexed by p' = g + 13."""
    N, h, w = lr.shape
    Hg, Wg = H + 27, W + 27

    def axis_tab(pl, n_img, length):
        tab = []
        for k in range(N):
            t = []
            for p in range(length):
                u = p + pl.n[k] - 13
                if u < 0:
                    t.append((0, -pl.n[k]))  # replicated LR row 0, subtracts Y[rho = -n_k]
                elif u <= n_img - 1 and u % f == 0:
                    t.append((u // f, p - 13))  # own position, rho = g
                else:
                    t.append((-1, 0))
            tab.append(t)
        return tab

    ty, tx = axis_tab(py, H, Hg), axis_tab(px, W, Wg)
    M = np.zeros((Hg, Wg))
    C = np.zeros((Hg, Wg), dtype=int)
    Mu = np.zeros((Hg, Wg))
    Cu = np.zeros((Hg, Wg), dtype=int)
    S2 = np.zeros((Hg, Wg))
    lists = {}
    for k in range(N):
        for p in range(Hg):
            iy, ry = ty[k][p]
            if iy < 0:
                continue
            for q in range(Wg):
                ix, rx = tx[k][q]
                if ix < 0:
                    continue
                v = lr[k, iy, ix]
                M[p, q] += v
                C[p, q] += 1
                if p < py.PB or q < px.PB:
                    lists.setdefault((p, q), []).append((ry, rx))
                if ry == p - 13 and rx == q - 13:
                    Mu[p, q] += v
                    Cu[p, q] += 1
                    S2[p, q] += v * v
    V = np.where(Cu > 1, S2 - Mu * Mu / np.maximum(Cu, 1), 0.0).sum()
    return M, C, Mu, Cu, V, lists


def forward_Y_dense(hr, psf, py, px):
    """Y in natural coordinates as a function Y(rho_y, rho_x) over rho in [-12 + 2, n + 10]."""
    H, W = hr.shape
    b = O.blur(hr, psf)
    bp = np.pad(b, NPAD, mode="edge")
    if py.zero and px.zero:
        def Y(ry, rx):
            return bp[np.clip(ry + 12, 0, H + 23), np.clip(rx + 12, 0, W + 23)]
        return Y
    c = prefilter_reflect(prefilter_reflect(bp, 0), 1)
    Hp, Wp = c.shape
    Yp = np.zeros((Hp - 3, Wp - 3))
    for a in range(4):
        for bb in range(4):
            Yp += py.wf[a] * px.wf[bb] * c[a:a + Hp - 3, bb:bb + Wp - 3]

    def Y(ry, rx):  # padded P = rho + E
        return Yp[ry + py.E, rx + px.E]
    return Y


def dense_G(Y, tabs, py, px, H, W, scale):
    M, C, Mu, Cu, V, lists = tabs
    Hg, Wg = M.shape
    G = np.zeros((Hg, Wg))
    sq = 0.0
    for p in range(py.RS, Hg):
        for q in range(px.RS, Wg):
            gy, gx = p - 13, q - 13
            if p < py.PB or q < px.PB:
                G[p, q] = M[p, q] - sum(Y(ry, rx) for ry, rx in lists.get((p, q), []))
                if Cu[p, q] > 0:
                    gu = Mu[p, q] - Cu[p, q] * Y(gy, gx)
                    sq += gu * gu / Cu[p, q]
            elif C[p, q] > 0:
                G[p, q] = M[p, q] - C[p, q] * Y(gy, gx)
                sq += G[p, q] ** 2 / C[p, q]
    G[:py.RS, :] = G[py.RS, :]
    G[:, :px.RS] = G[:, px.RS:px.RS + 1]
    return G, (sq + V) * scale


def backward_dense(G, psf, py, px, H, W):
    Hg, Wg = G.shape
    Hp, Wp = Hg - 3, Wg - 3
    if py.zero and px.zero:
        cc = G[1:1 + Hp, 1:1 + Wp]
    else:
        v = np.zeros((Hp, Wp))
        for a in range(4):
            for bb in range(4):
                v += py.wb[a] * px.wb[bb] * G[a:a + Hp, bb:bb + Wp]
        cc = prefilter_reflect(prefilter_reflect(v, 0), 1)
    d = cc[NPAD:NPAD + H, NPAD:NPAD + W]
    return O.blur(d, psf[::-1, ::-1])


def dense_iteration(hr, lr, shifts, psf, f, step, tabs=None):
    N, h, w = lr.shape
    H, W = hr.shape
    py, px = AxisPlan(shifts, 0, f), AxisPlan(shifts, 1, f)
    if tabs is None:
        tabs = build_tables(lr, py, px, f, H, W)
    Y = forward_Y_dense(hr, psf, py, px)
    G, err = dense_G(Y, tabs, py, px, H, W, 1.0 / (h * w) / N)
    corr = backward_dense(G, psf, py, px, H, W)
    return np.clip(hr + step * corr / N, 0.0, 255.0), err, G


# --------------------------------------------------------------------------------------------
# layer 2: the per-axis chains, block by block
# --------------------------------------------------------------------------------------------
class Chain1D:
    """Operator chains along axis 0 of an [n, m] array, n = nb * 64, evaluated as the kernel's waves do: every 64-sample
    block runs its recursions locally from a zero (or closed-form) state and adds the neighbour's carry afterwards over its
    first / last `fix` samples.  fix = 64 makes the carries exact (float64 check); the float kernel uses 16."""

    def __init__(self, pl, n, fix=64, BS=64, dtype=np.float64):
        self.pl, self.n, self.fix, self.BS, self.dt = pl, n, fix, BS, dtype
        assert n % BS == 0 and not pl.zero
        self.nb = n // BS
        self.ex = max(int(pl.n.max()), 0)     # extra rows above the block grid
        self.zp = (Z ** np.arange(1, BS + 1)).astype(dtype)  # z^1 .. z^BS
        self.z = dtype(Z)

    def _prefilter_blocks(self, v, cp_in, c_below_fn):
        """v [n, m] input; cp_in [m] = c+ of the sample just above v[0]; c_below_fn(cplus_last) = coefficient just below
        v[n-1].  Returns (c [n, m], c_below)."""
        BS, nb, z, dt = self.BS, self.nb, self.z, self.dt
        n, m = v.shape
        cp = np.empty_like(v)
        for s in range(nb):   # local causal recursions
            st = cp_in.astype(dt) if s == 0 else np.zeros(m, dt)
            for i in range(BS):
                st = dt(6.0) * v[s * BS + i] + z * st
                cp[s * BS + i] = st
        ends = [cp[s * BS + BS - 1].copy() for s in range(nb)]
        for s in range(1, nb):   # carry of the block above = its local end (z^64 = 0 in either precision)
            for i in range(min(self.fix, BS)):
                cp[s * BS + i] += self.zp[i] * ends[s - 1]
        c_below = c_below_fn(cp[n - 1])
        c = np.empty_like(v)
        for s in range(nb):   # local anticausal recursions
            nxt = c_below.astype(dt) if s == nb - 1 else np.zeros(m, dt)
            for i in range(BS - 1, -1, -1):
                nxt = z * (nxt - cp[s * BS + i])
                c[s * BS + i] = nxt
        tops = [c[s * BS].copy() for s in range(nb)]
        for s in range(nb - 1):
            for i in range(min(self.fix, BS)):
                c[s * BS + BS - 1 - i] += self.zp[i] * tops[s + 1]
        return c, c_below

    def forward(self, x, k7):
        """x [n, m] -> Y [ex + n, m], rows rho = -ex .. n-1:  Y[rho] = sum_a wf[a] c[rho - 2 + a], c = P(pad12(corr7(x)))."""
        pl, n, ex, z, dt = self.pl, self.n, self.ex, self.z, self.dt
        wf = pl.wf.astype(dt)
        b = corr7_1d(x, k7.astype(dt), 0).astype(dt)
        S_top = dt(6.0) * b[0] / (dt(1.0) - z)         # c+ inside the constant top pad (steady state)
        S_bot = dt(6.0) * b[n - 1] / (dt(1.0) - z)

        def below(cp_last):  # coefficient of the first bottom-pad sample: 12 constant samples, then the reflect end
            return -z * S_bot / (dt(1.0) - z) - z * z * (cp_last - S_bot) / (dt(1.0) - z * z)
        c, c_below = self._prefilter_blocks(b, S_top, below)
        ntop = ex + 2                                  # pad coefficients c[-1], c[-2], ...: c[i] = z (c[i+1] - S_top)
        ctop = np.empty((ntop,) + b.shape[1:], dt)
        nxt = c[0]
        for j in range(ntop):
            nxt = z * (nxt - S_top)
            ctop[ntop - 1 - j] = nxt
        cext = np.concatenate([ctop, c, c_below[None]])   # cext[i + ntop] = c[i]
        Y = np.empty((ex + n,) + b.shape[1:], dt)
        for r in range(-ex, n):
            Y[r + ex] = wf[0] * cext[r - 2 + ntop] + wf[1] * cext[r - 1 + ntop] + wf[2] * cext[r + ntop] + wf[3] * cext[r + 1 + ntop]
        return Y

    def backward(self, Gx, k7t):
        """Gx [ex + n, m], rows g = -ex .. n-1 (rows above are replicas of row -ex, rows >= n are zero)
        -> corr [n, m] = corr7_t( crop P( FIR_b G ) )."""
        pl, n, ex, z, dt = self.pl, self.n, self.ex, self.z, self.dt
        wb = pl.wb.astype(dt)

        def G(g):
            if g >= n:
                return np.zeros(Gx.shape[1:], dt)
            return Gx[max(g, -ex) + ex]
        v = np.empty((n,) + Gx.shape[1:], dt)
        for t in range(n):
            v[t] = wb[0] * G(t - 1) + wb[1] * G(t) + wb[2] * G(t + 1) + wb[3] * G(t + 2)
        # top pad: steady state of the constant run, then the ex + 1 pad samples whose FIR window reaches real rows
        st = dt(6.0) * G(-ex) / (dt(1.0) - z)
        for t in range(-ex - 1, 0):
            vt = wb[0] * G(t - 1) + wb[1] * G(t) + wb[2] * G(t + 1) + wb[3] * G(t + 2)
            st = dt(6.0) * vt + z * st
        vn = wb[0] * G(n - 1)                            # v[n]: the one bottom-pad sample that can be non-zero

        def below(cp_last):
            return -z * (dt(6.0) * vn + z * cp_last) / (dt(1.0) - z * z)
        c, _ = self._prefilter_blocks(v, st, below)
        return corr7_1d(c, k7t.astype(dt), 0).astype(dt)


def patch_iteration(hr, lr, shifts, psf, f, step, tabs=None, fix=64, dtype=np.float64):
    """The kernel's data flow: V-fwd chain (columns), H-fwd chain (rows, extra rows included), dense G step, H-bwd, V-bwd."""
    N, h, w = lr.shape
    H, W = hr.shape
    py, px = AxisPlan(shifts, 0, f), AxisPlan(shifts, 1, f)
    if tabs is None:
        tabs = build_tables(lr, py, px, f, H, W)
    cy, cx = sep_kernel(psf)
    cyt, cxt = sep_kernel(psf[::-1, ::-1])
    chy, chx = Chain1D(py, H, fix, dtype=dtype), Chain1D(px, W, fix, dtype=dtype)
    exy, exx = chy.ex, chx.ex
    hr = hr.astype(dtype)
    Yv = chy.forward(hr, cy)                    # [exy + H, W]
    Yf = chx.forward(Yv.T.copy(), cx).T          # [exy + H, exx + W]

    def Y(ry, rx):
        return float(Yf[ry + exy, rx + exx])
    G, err = dense_G(Y, tabs, py, px, H, W, 1.0 / (h * w) / N)
    # natural-coordinate window of G the chains read: rows g = -exy .. H-1 <-> p' = g + 13
    Gx = G[13 - exy:13 + H, 13 - exx:13 + W].astype(dtype)
    assert np.all(G[13 + H:, :] == 0) and np.all(G[:, 13 + W:] == 0)
    Hb = chx.backward(Gx.T.copy(), cxt).T        # [exy + H, W]
    corr = chy.backward(Hb, cyt)                 # [H, W]
    sn = dtype(step) / dtype(N)
    return np.clip(hr + corr * sn, dtype(0), dtype(255)), err


def main():
    f, N = 4, 16
    h = w = 64
    H, W = h * f, w * f
    shifts = synth.phase_shifts(f)
    psf = synth.gaussian_psf()
    truth = synth.truth_image(H, W)
    lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]))
    hr0 = O.shift_and_add(list(lr), shifts, f)
    ref, errs = O.ibp(list(lr), shifts, psf, hr0, f, 1, 0.5)
    out, err, G = dense_iteration(hr0, lr, shifts, psf, f, 0.5)
    print("dense vs oracle: max|d| = %.3e   mse %.12g vs %.12g" % (np.abs(out - ref).max(), err, errs[0]))
    py, px = AxisPlan(shifts, 0, f), AxisPlan(shifts, 1, f)
    tabs = build_tables(lr, py, px, f, H, W)
    o2, e2 = patch_iteration(hr0, lr, shifts, psf, f, 0.5, tabs)
    print("blocked (fix 64, f64) vs oracle: max|d| = %.3e   mse rel %.3e" % (np.abs(o2 - ref).max(), abs(e2 / errs[0] - 1)))
    for fix in (24, 16, 12):
        o3, e3 = patch_iteration(hr0, lr, shifts, psf, f, 0.5, tabs, fix=fix)
        print("blocked (fix %d, f64) vs oracle: max|d| = %.3e" % (fix, np.abs(o3 - ref).max()))
    o4, e4 = patch_iteration(hr0, lr, shifts, psf, f, 0.5, tabs, fix=16, dtype=np.float32)
    print("blocked (fix 16, f32) vs oracle: max|d| = %.3e   mse rel %.3e" % (np.abs(o4 - ref).max(), abs(e4 / errs[0] - 1)))


if __name__ == "__main__":
    main()
