"""numpy statement of the multigrid operators on an i-slab (planes [ig0, ig0+ni) of an N^3 level), used only
by the CPU multi-process test of the slab schedule.  Each expression keeps the reference's association, so
with float64 ufuncs (no fusion) every value is bit-identical to the oracle's.  Test infrastructure only."""
import numpy as np


def colour_pass(u, d, h, colour, ig0, N):
    """mg_3d.h:438-443 on the local planes 1..ni-2 that are interior globally; the two end planes act as fixed
    (possibly stale) data, exactly like the HIP sweep on a slab."""
    ni = u.shape[0]
    hSq = h * h
    sixth = 1.0 / 6
    lo = max(1, 1 - ig0)
    hi = min(ni - 2, N - 2 - ig0)
    if hi < lo:
        return
    c = u[lo:hi + 1, 1:-1, 1:-1]
    s = u[lo - 1:hi, 1:-1, 1:-1] + u[lo + 1:hi + 2, 1:-1, 1:-1]
    s = s + u[lo:hi + 1, 0:-2, 1:-1]
    s = s + u[lo:hi + 1, 2:, 1:-1]
    s = s + u[lo:hi + 1, 1:-1, 0:-2]
    s = s + u[lo:hi + 1, 1:-1, 2:]
    s = s - hSq * d[lo:hi + 1, 1:-1, 1:-1]
    new = sixth * s
    ii = (np.arange(lo, hi + 1) + ig0)[:, None, None]
    jj = np.arange(1, N - 1)[None, :, None]
    kk = np.arange(1, N - 1)[None, None, :]
    mask = ((ii + jj + kk) & 1) == colour
    c[mask] = new[mask]


def smooth(u, d, h, iters, post, ig0, N):
    for _ in range(iters):
        colour_pass(u, d, h, 0 if post else 1, ig0, N)
        colour_pass(u, d, h, 1 if post else 0, ig0, N)


def residual(u, d, h, r, ig0, N, acc_lo=None, acc_hi=None):
    """mg_3d.h:819-821; r written on interior points of local planes 1..ni-2; returns sum diff^2 over the local
    planes [acc_lo, acc_hi)."""
    ni = u.shape[0]
    invHsq = 1.0 / (h * h)
    lo = max(1, 1 - ig0)
    hi = min(ni - 2, N - 2 - ig0)
    if hi < lo:
        return 0.0
    s = u[lo - 1:hi, 1:-1, 1:-1] + u[lo + 1:hi + 2, 1:-1, 1:-1]
    s = s + u[lo:hi + 1, 0:-2, 1:-1]
    s = s + u[lo:hi + 1, 2:, 1:-1]
    s = s + u[lo:hi + 1, 1:-1, 0:-2]
    s = s + u[lo:hi + 1, 1:-1, 2:]
    s = s - 6 * u[lo:hi + 1, 1:-1, 1:-1]
    diff = d[lo:hi + 1, 1:-1, 1:-1] - invHsq * s
    if r is not None:
        r[lo:hi + 1, 1:-1, 1:-1] = diff
    a = lo if acc_lo is None else max(lo, acc_lo)
    b = hi + 1 if acc_hi is None else min(hi + 1, acc_hi)
    return float((diff[a - lo:b - lo] ** 2).sum()) if b > a else 0.0


def restrict_planes(r, igf0, Nf, dc, igc0, Nc, ic_lo, ic_hi):
    """mg_3d.h:844-998 for the local coarse planes [ic_lo, ic_hi)."""
    w1 = (0.25, 0.5, 0.25)
    for icl in range(ic_lo, ic_hi):
        ic = igc0 + icl
        fi = 2 * ic - igf0
        inj = r[fi, 0::2, 0::2]
        if ic == 0 or ic == Nc - 1:
            dc[icl] = inj
            continue
        out = inj.copy()  # faces in j, k: injection
        val = np.zeros((Nc - 2, Nc - 2))
        for ti in range(3):
            for tj in range(3):
                for tk in range(3):
                    w = w1[ti] * w1[tj] * w1[tk]
                    val = val + r[fi - 1 + ti, 1 + tj:Nf - 2 + tj:2, 1 + tk:Nf - 2 + tk:2] * w
        out[1:-1, 1:-1] = val
        dc[icl] = out


def prolong_planes(ec, igc0, Nc, ef, igf0, Nf, if_lo, if_hi):
    """mg_3d.h:1000-1145 for the local fine planes [if_lo, if_hi): ef += P(ec), parents in the reference's order."""
    for il in range(if_lo, if_hi):
        ig = igf0 + il
        oi = ig & 1
        lo = (ig - oi) // 2 - igc0
        A = ec[lo]
        B = ec[lo + 1] if oi else None
        add = np.empty((Nf, Nf))
        if not oi:
            add[0::2, 0::2] = A
            add[1::2, 0::2] = (A[:-1, :] + A[1:, :]) * 0.5
            add[0::2, 1::2] = (A[:, :-1] + A[:, 1:]) * 0.5
            add[1::2, 1::2] = (((A[:-1, :-1] + A[1:, :-1]) + A[:-1, 1:]) + A[1:, 1:]) * 0.25  # :1064-1067
        else:
            add[0::2, 0::2] = (A + B) * 0.5
            add[1::2, 0::2] = (((A[:-1, :] + A[1:, :]) + B[:-1, :]) + B[1:, :]) * 0.25    # k even: :1085-1088
            add[0::2, 1::2] = (((A[:, :-1] + B[:, :-1]) + A[:, 1:]) + B[:, 1:]) * 0.25    # j even: :1075-1078
            t = A[:-1, :-1] + A[:-1, 1:]
            t = t + A[1:, :-1]
            t = t + A[1:, 1:]
            t = t + B[:-1, :-1]
            t = t + B[:-1, 1:]
            t = t + B[1:, :-1]
            t = t + B[1:, 1:]
            add[1::2, 1::2] = t * 0.125                                                      # :1028-1048
        ef[il] += add
