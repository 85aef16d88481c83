"""TEST INFRASTRUCTURE (checker only): numpy restatement of the per-DOF expression interpreter of
atomsmm_amd/csrc/expr.hip -- the same postfix programs, the same Philox-4x32-10 stream -- used by tests to check the GPU
kernel, never by the product.  Philox follows Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3"
(SC'11); the Gaussian is Box-Muller on two 53-bit uniforms."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniforms(ndof, occurrence, seed, counter):
    dof = np.arange(ndof, dtype=np.uint64)
    occ = np.full(ndof, occurrence, dtype=np.uint64)
    r = philox4x32_10(dof, occ, np.full(ndof, counter & 0xFFFFFFFF, dtype=np.uint64),
                      np.full(ndof, (counter >> 32) & 0xFFFFFFFF, dtype=np.uint64), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = ((r[0] << np.uint64(21)) ^ (r[1] >> np.uint64(11))) & np.uint64(0x1FFFFFFFFFFFFF)
    b = ((r[2] << np.uint64(21)) ^ (r[3] >> np.uint64(11))) & np.uint64(0x1FFFFFFFFFFFFF)
    return (a.astype(np.float64) + 0.5) / 9007199254740992.0, (b.astype(np.float64) + 0.5) / 9007199254740992.0


def run(code, consts, globals_, bufs, mass, seed, counter):
    """Values of the program for every DOF: bufs = {slot: array [n][3]}, mass [n]."""
    from scipy.special import erf, erfc
    ndof = 3 * len(mass)
    st, loc = [], {}
    one = lambda f: st.append(f(st.pop()))            # noqa: E731
    for word in code:
        op, arg = word & 0xff, word >> 8
        if op == 0:
            st.append(np.full(ndof, consts[arg]))
        elif op == 1:
            st.append(np.full(ndof, globals_[arg]))
        elif op == 2:
            st.append(np.asarray(bufs[arg], dtype=np.float64).reshape(-1).copy())
        elif op == 3:
            st.append(np.repeat(np.asarray(mass, dtype=np.float64), 3))
        elif op == 4:
            u1, u2 = uniforms(ndof, 0, seed, counter)
            st.append(np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2))
        elif op == 5:
            st.append(uniforms(ndof, 1, seed, counter)[0])
        elif op == 6:
            st.append(loc[arg].copy())
        elif op == 7:
            loc[arg] = st.pop()
        elif op in (10, 11, 12, 13, 15, 39, 40, 42):
            b = st.pop()
            a = st.pop()
            st.append({10: np.add, 11: np.subtract, 12: np.multiply, 13: np.divide, 15: np.power, 39: np.minimum,
                       40: np.maximum, 42: np.arctan2}[op](a, b))
        elif op == 14:
            one(np.negative)
        elif op == 16:
            one(lambda a: a ** arg if arg >= 0 else 1.0 / a ** (-arg))
        elif op == 41:
            c = st.pop(); b = st.pop(); a = st.pop()
            st.append(np.where(a != 0.0, b, c))
        else:
            one({20: np.sqrt, 21: np.exp, 22: np.log, 23: np.sin, 24: np.cos, 25: np.tan, 26: np.arcsin, 27: np.arccos,
                 28: np.arctan, 29: np.sinh, 30: np.cosh, 31: np.tanh, 32: erf, 33: erfc, 34: np.abs, 35: np.floor,
                 36: np.ceil, 37: lambda a: (a >= 0).astype(float), 38: lambda a: (a == 0).astype(float)}[op])
    assert len(st) == 1
    return st[0].reshape(-1, 3)
