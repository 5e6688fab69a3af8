"""BN254 (alt_bn128) optimal-ate pairing in plain Python integers — the oracle-side verifier of the decider's Groth16 proof
(vimz_amd/csrc/groth16.hip; the reference's decider is `DeciderEth<..., Groth16<Bn254>, ...>`, vimz/src/sonobe_backend/decider.rs:13-21,
checked on chain by the pairing precompile behind contracts/*Verifier.sol).  Test infrastructure only: nothing of the product imports this.

Construction (the textbook one, as the EVM precompile specifies it): Fq12 = Fq[w] / (w^12 - 18 w^6 + 82), Fq2 = Fq[u] / (u^2 + 1) embedded by
u = w^6 - 9; G2 points on the twist y^2 = x^3 + 3/(9 + u) are mapped to E(Fq12) by (x, y) -> (x w^2, y w^3); Miller loop over 6t + 2 with the
two Frobenius corrections; final exponentiation by (q^12 - 1) / r.  Slow and simple on purpose: every division is a polynomial extended Euclid."""

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583      # base field
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617      # group order
ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE_LOOP_COUNT = 63
G1 = (1, 2)
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))
MOD12 = [82, 0, 0, 0, 0, 0, -18 % Q, 0, 0, 0, 0, 0]      # w^12 = 18 w^6 - 82


# ---- Fq12 as polynomials of degree < 12 over Fq ---------------------------------------------------------------------------------------
def f12(coeffs):
    return tuple(c % Q for c in coeffs)


F12_ONE = f12([1] + [0] * 11)
F12_ZERO = f12([0] * 12)
W = f12([0, 1] + [0] * 10)


def f12_add(a, b):
    return tuple((x + y) % Q for x, y in zip(a, b))


def f12_sub(a, b):
    return tuple((x - y) % Q for x, y in zip(a, b))


def f12_neg(a):
    return tuple(-x % Q for x in a)


def f12_scale(a, k):
    return tuple(x * k % Q for x in a)


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):      # w^k = w^(k-12) (18 w^6 - 82)
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return tuple(x % Q for x in t[:12])


def _poly_deg(p):
    d = len(p) - 1
    while d > 0 and p[d] == 0:
        d -= 1
    return d


def _poly_divmod(a, b):
    """Quotient and remainder of polynomials over Fq (coefficient lists, lowest degree first; b != 0)."""
    a = [x % Q for x in a]
    da, db = _poly_deg(a), _poly_deg(b)
    if da < db or (da == 0 and a[0] == 0):
        return [0], a
    inv = pow(b[db], Q - 2, Q)
    quo = [0] * (da - db + 1)
    for i in range(da - db, -1, -1):
        c = a[db + i] * inv % Q
        quo[i] = c
        if c:
            for k in range(db + 1):
                a[k + i] = (a[k + i] - c * b[k]) % Q
    return quo, a[:max(1, db)]


def _poly_mul(a, b):
    o = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                o[i + j] = (o[i + j] + x * y) % Q
    return o


def _poly_sub(a, b):
    n = max(len(a), len(b))
    return [((a[k] if k < len(a) else 0) - (b[k] if k < len(b) else 0)) % Q for k in range(n)]


def f12_inv(a):
    """1 / a in Fq12 by the extended Euclidean algorithm on polynomials (s_k · a = r_k modulo the field's polynomial)."""
    if all(x == 0 for x in a):
        raise ZeroDivisionError("Fq12 inverse of zero")
    r0, s0 = [x % Q for x in MOD12] + [1], [0]
    r1, s1 = list(a), [1]
    while _poly_deg(r1) > 0:
        quo, rem = _poly_divmod(r0, r1)
        r0, r1 = r1, rem
        s0, s1 = s1, _poly_sub(s0, _poly_mul(quo, s1))
    c = pow(r1[0], Q - 2, Q)
    out = [x * c % Q for x in s1] + [0] * 12
    # (deg s1 <= 11: no reduction needed)
    assert all(x == 0 for x in out[12:])
    return tuple(out[:12])


def f12_div(a, b):
    return f12_mul(a, f12_inv(b))


def f12_pow(a, e):
    r, base = F12_ONE, a
    while e:
        if e & 1:
            r = f12_mul(r, base)
        base = f12_mul(base, base)
        e >>= 1
    return r


# ---- curve arithmetic over Fq12 (affine; None = infinity) ---------------------------------------------------------------------------------
def _double(p):
    x, y = p
    m = f12_div(f12_scale(f12_mul(x, x), 3), f12_scale(y, 2))
    nx = f12_sub(f12_mul(m, m), f12_scale(x, 2))
    return nx, f12_sub(f12_mul(m, f12_sub(x, nx)), y)


def _add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    m = f12_div(f12_sub(y2, y1), f12_sub(x2, x1))
    nx = f12_sub(f12_sub(f12_mul(m, m), x1), x2)
    return nx, f12_sub(f12_mul(m, f12_sub(x1, nx)), y1)


def _linefunc(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f12_div(f12_sub(y2, y1), f12_sub(x2, x1))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_div(f12_scale(f12_mul(x1, x1), 3), f12_scale(y1, 2))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


def twist(q2):
    """A G2 point ((x0, x1), (y0, y1)) (x = x0 + x1 u) as a point of E(Fq12)."""
    (x0, x1), (y0, y1) = q2
    nx = f12([x0 - 9 * x1] + [0] * 5 + [x1] + [0] * 5)
    ny = f12([y0 - 9 * y1] + [0] * 5 + [y1] + [0] * 5)
    w2 = f12_mul(W, W)
    return f12_mul(nx, w2), f12_mul(ny, f12_mul(w2, W))


def cast_g1(p1):
    return f12([p1[0]] + [0] * 11), f12([p1[1]] + [0] * 11)


def miller_loop(q2, p1):
    """The Miller function of the optimal ate pairing for Q in G2 (twist coordinates) and P in G1 (affine); 1 when either is infinity."""
    if q2 is None or p1 is None:
        return F12_ONE
    Qt, P = twist(q2), cast_g1(p1)
    Rp, f = Qt, F12_ONE
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(Rp, Rp, P))
        Rp = _double(Rp)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, _linefunc(Rp, Qt, P))
            Rp = _add(Rp, Qt)
    Q1 = (f12_pow(Qt[0], Q), f12_pow(Qt[1], Q))
    nQ2 = (f12_pow(Q1[0], Q), f12_neg(f12_pow(Q1[1], Q)))
    f = f12_mul(f, _linefunc(Rp, Q1, P))
    Rp = _add(Rp, Q1)
    return f12_mul(f, _linefunc(Rp, nQ2, P))


def final_exponentiate(f):
    return f12_pow(f, (Q ** 12 - 1) // R)


def pairing(q2, p1):
    return final_exponentiate(miller_loop(q2, p1))


def pairing_product_is_one(pairs):
    """prod e(P_k, Q_k) == 1 for pairs [(G1 point, G2 point)] — what the EVM's pairing precompile (0x08) answers: one Miller loop per
    pair, ONE final exponentiation."""
    f = F12_ONE
    for p1, q2 in pairs:
        f = f12_mul(f, miller_loop(q2, p1))
    return final_exponentiate(f) == F12_ONE


# ---- G1 / G2 in their own coordinates (checks and small multiples for the tests) ----------------------------------------------------------
def g1_on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - 3) % Q == 0


def g1_neg(p):
    return None if p is None else (p[0], -p[1] % Q)


def g1_add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    if p1[0] == p2[0]:
        if (p1[1] + p2[1]) % Q == 0:
            return None
        m = 3 * p1[0] * p1[0] * pow(2 * p1[1], Q - 2, Q) % Q
    else:
        m = (p2[1] - p1[1]) * pow(p2[0] - p1[0], Q - 2, Q) % Q
    x = (m * m - p1[0] - p2[0]) % Q
    return x, (m * (p1[0] - x) - p1[1]) % Q


def g1_mul(p, k):
    r, k = None, k % R
    while k:
        if k & 1:
            r = g1_add(r, p)
        p = g1_add(p, p)
        k >>= 1
    return r


def _f2_mul(a, b):
    return (a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q


def _f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], Q - 2, Q)
    return a[0] * d % Q, -a[1] * d % Q


B2 = _f2_mul((3, 0), _f2_inv((9, 1)))      # 3 / (9 + u)


def g2_on_curve(p):
    if p is None:
        return True
    x, y = p
    x3 = _f2_mul(_f2_mul(x, x), x)
    y2 = _f2_mul(y, y)
    return ((y2[0] - x3[0] - B2[0]) % Q, (y2[1] - x3[1] - B2[1]) % Q) == (0, 0)


def g2_add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if ((y1[0] + y2[0]) % Q, (y1[1] + y2[1]) % Q) == (0, 0):
            return None
        xx = _f2_mul(x1, x1)
        m = _f2_mul((3 * xx[0] % Q, 3 * xx[1] % Q), _f2_inv((2 * y1[0] % Q, 2 * y1[1] % Q)))
    else:
        m = _f2_mul(((y2[0] - y1[0]) % Q, (y2[1] - y1[1]) % Q), _f2_inv(((x2[0] - x1[0]) % Q, (x2[1] - x1[1]) % Q)))
    mm = _f2_mul(m, m)
    x = ((mm[0] - x1[0] - x2[0]) % Q, (mm[1] - x1[1] - x2[1]) % Q)
    t = _f2_mul(m, ((x1[0] - x[0]) % Q, (x1[1] - x[1]) % Q))
    return x, ((t[0] - y1[0]) % Q, (t[1] - y1[1]) % Q)


def g2_mul(p, k):
    r, k = None, k % R
    while k:
        if k & 1:
            r = g2_add(r, p)
        p = g2_add(p, p)
        k >>= 1
    return r


def groth16_verify(vk, public_inputs, proof):
    """e(A, B) == e(alpha, beta) · e(sum_i x_i IC_i, gamma) · e(C, delta)   with x_0 = 1.
    vk: {"alpha": G1, "beta": G2, "gamma": G2, "delta": G2, "ic": [G1 ...]};  proof: (A: G1, B: G2, C: G1).  Every point must be on its curve."""
    A, B, C = proof
    if len(public_inputs) + 1 != len(vk["ic"]):
        return False
    pts1 = [A, C, vk["alpha"]] + list(vk["ic"])
    pts2 = [B, vk["beta"], vk["gamma"], vk["delta"]]
    if not all(g1_on_curve(p) for p in pts1) or not all(g2_on_curve(p) for p in pts2):
        return False
    acc = vk["ic"][0]
    for x, ic in zip(public_inputs, vk["ic"][1:]):
        acc = g1_add(acc, g1_mul(ic, x))
    return pairing_product_is_one([(A, B), (g1_neg(vk["alpha"]), vk["beta"]), (g1_neg(acc), vk["gamma"]), (g1_neg(C), vk["delta"])])
