#!/usr/bin/env python3
"""Generates bls-verify-gadget_amd/csrc/team_tables.hpp: the op tables of the 6-lanes-per-instance pairing kernel.

An Fp12 value is DISTRIBUTED over the six lanes of a team: lane j owns the Fp2 coefficient j of
(c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2). A tower operation (Fp12 mul / square / cyclotomic square / mul_by_014 ...)
is a set of independent Fp2 products ("tasks", scheduled 6 per round) whose operands are small linear combinations of
coefficients published in LDS, followed by a linear recombination per output coefficient. This script expands the
formulas of tower.hpp (= ark-r1cs-std's allocation order) symbolically and emits, per op:
  task[round][lane] = {kind, result slot, witness offset, operand A, operand B}     out[lane] = linear combination
A linear combination is  (sum Lp - sum Ln) + xi * (sum Mp - sum Mn)  over LDS slots, xi = 1 + u.
Witness offsets follow the order in which the single-lane code (tower.hpp) emits them; tests pin both against the oracle.
"""
import os
import sys

IN0, IN1, P0, X0 = 0, 6, 12, 30
N_P = 18
XS0, XS1, XH0, XH1, XYC, XYV, XPX = range(X0, X0 + 7)
N_SLOTS = X0 + 7
K_NONE, K3, K3V, K2, K2V, K2B, K1E, K2S = range(8)
KIND_NAMES = ["TK_NONE", "TK_K3", "TK_K3V", "TK_K2", "TK_K2V", "TK_K2B", "TK_K1E", "TK_K2S"]
KIND_WITNESSES = {K3: 3, K3V: 0, K2: 2, K2V: 0, K2B: 2, K1E: 1, K2S: 2}
MAX_TERMS = 16


class LC:
    """sum over slots of (a + b*xi) * slot"""

    def __init__(self, d=None):
        self.d = {k: v for k, v in (d or {}).items() if v != (0, 0)}

    @staticmethod
    def slot(s):
        return LC({s: (1, 0)})

    def __add__(self, o):
        d = dict(self.d)
        for k, (a, b) in o.d.items():
            a0, b0 = d.get(k, (0, 0))
            d[k] = (a0 + a, b0 + b)
        return LC(d)

    def __neg__(self):
        return LC({k: (-a, -b) for k, (a, b) in self.d.items()})

    def __sub__(self, o):
        return self + (-o)

    def dbl(self):
        return self + self

    def xi(self):
        for k, (a, b) in self.d.items():
            if b:
                raise ValueError("xi^2 is not representable: split the op")
        return LC({k: (0, a) for k, (a, b) in self.d.items()})

    def mscale(self):
        """common factor of the xi part that the device applies after summing: 12 (= 3b of y^2 = x^3 + 4 xi) or 24"""
        import math
        g = 0
        for a, b in self.d.values():
            g = math.gcd(g, abs(b))
        return 24 if g and g % 24 == 0 else (12 if g and g % 12 == 0 else 1)

    def lists(self):
        lp, ln, mp, mn = [], [], [], []
        sc = self.mscale()
        for k in sorted(self.d):
            a, b = self.d[k]
            (lp if a > 0 else ln).extend([k] * abs(a))
            (mp if b > 0 else mn).extend([k] * (abs(b) // sc))
        if len(lp) + len(ln) + len(mp) + len(mn) > MAX_TERMS:
            raise ValueError("linear combination too long: %d" % (len(lp) + len(ln) + len(mp) + len(mn)))
        return lp, ln, mp, mn

    def depends_on_products(self):
        return any(P0 <= k < P0 + N_P for k in self.d)


ZERO = LC()


class Op:
    def __init__(self, name, value_only=False):
        self.name = name
        self.value_only = value_only  # every product as the value-only kind (no witness): the native pairing of blsw_verify_batch
        self.tasks = []  # dicts in emission order
        self.woff = 0
        self.np = 0
        self.out = [ZERO] * 6

    def _task(self, kind, a, b, want_result=True):
        dst = 0xFF
        if want_result:
            assert self.np < N_P, "out of product slots"
            dst = P0 + self.np
            self.np += 1
        self.tasks.append({"kind": kind, "a": a, "b": b, "dst": dst, "woff": self.woff})
        self.woff += KIND_WITNESSES[kind]
        return LC.slot(dst) if want_result else None

    def mul3(self, a, b):
        return self._task(K3V if self.value_only else K3, a, b)

    def mul2(self, a, y, witness):  # Fp2 x (y, 0)
        return self._task(K2 if witness else K2V, a, y)

    def mul2b(self, a, y):
        return self._task(K2B, a, y)

    def muleq(self, a, b):
        self._task(K1E, a, b, want_result=False)

    def sqr2(self, a):  # Fp2 square: a0*a1, (a0 - a1)(a0 + a1)
        return self._task(K2S, a, ZERO)

    def schedule(self):
        """rounds of <= 6 tasks of one kind; a task that reads a product goes after the round that produced it"""
        level = {}
        for i, t in enumerate(self.tasks):
            lv = 0
            for lc in (t["a"], t["b"]):
                for k in lc.d:
                    if P0 <= k < P0 + N_P:
                        prod = next(ix for ix, u in enumerate(self.tasks) if u["dst"] == k)
                        lv = max(lv, level[prod] + 1)
            level[i] = lv
        # kinds that run the same instruction stream share rounds: the two-product kinds (K2, K2V, K2B) and the Karatsuba kinds
        classes = ((K2B, K2, K2V), (K3, K3V, K2S), (K1E,))
        rounds = []
        for lv in sorted(set(level.values())):
            for cls in classes:
                ids = [i for i in range(len(self.tasks)) if level[i] == lv and self.tasks[i]["kind"] in cls]
                for c in range(0, len(ids), 6):
                    rounds.append(ids[c:c + 6])
        return rounds


# ---- tower formulas on LC-valued coefficients (tower.hpp)
def fp6_add(a, b):
    return [x + y for x, y in zip(a, b)]


def fp6_sub(a, b):
    return [x - y for x, y in zip(a, b)]


def fp6_mul_v(a):
    return [a[2].xi(), a[0], a[1]]


def fp6_mul_w(op, a, b):
    v0 = op.mul3(a[0], b[0])
    v1 = op.mul3(a[1], b[1])
    v2 = op.mul3(a[2], b[2])
    t0 = op.mul3(a[1] + a[2], b[1] + b[2])
    c0 = (t0 - v1 - v2).xi() + v0
    t1 = op.mul3(a[0] + a[1], b[0] + b[1])
    c1 = t1 - v0 - v1 + v2.xi()
    t2 = op.mul3(a[0] + a[2], b[0] + b[2])
    c2 = t2 - v0 + v1 - v2
    return [c0, c1, c2]


def fp6_mul_equals_w(op, a, b):
    op.mul3(a[0], b[0])
    op.mul3(a[1], b[1])
    op.mul3(a[2], b[2])
    op.muleq((a[1] + a[2]).xi(), b[1] + b[2])
    op.muleq(a[0] + a[1], b[0] + b[1])
    op.muleq(a[0] + a[2], b[0] + b[2])


def fp6_mul_by_c0_c1_0_w(op, a, c0, c1):
    v0 = op.mul3(a[0], c0)
    v1 = op.mul3(a[1], c1)
    t0 = op.mul3(a[1] + a[2], c1)
    r0 = (t0 - v1).xi() + v0
    t1 = op.mul3(a[0] + a[1], c0 + c1)
    r1 = t1 - v0 - v1
    t2 = op.mul3(a[0] + a[2], c0)
    r2 = t2 - v0 + v1
    return [r0, r1, r2]


def fp6_mul_by_0_y_0(op, a, y, witness):
    v1 = op.mul2(a[1], y, witness)
    t0 = op.mul2(a[1] + a[2], y, witness)
    r0 = (t0 - v1).xi()
    t1 = op.mul2(a[0] + a[1], y, witness)
    r1 = t1 - v1
    return [r0, r1, v1]


def fp12_mul_by_014_w(op, f, c0, c1, y, yvar):
    v0 = fp6_mul_by_c0_c1_0_w(op, f[0], c0, c1)
    v1 = fp6_mul_by_0_y_0(op, f[1], y, yvar)
    new_c0 = fp6_add(fp6_mul_v(v1), v0)
    t = fp6_mul_by_c0_c1_0_w(op, fp6_add(f[0], f[1]), c0, c1 + y)  # c1 + d1, d1 = (y, 0)
    new_c1 = fp6_sub(fp6_sub(t, v0), v1)
    return [new_c0, new_c1]


def fp12_mul_by_014_general(op, f, c0, c1, c4):
    """f * (c0 + c1 v + c4 v w) with THREE general Fp2 coefficients (ark-ff Fp12::mul_by_014): 13 Fp2 products, value only"""
    aa = fp6_mul_by_c0_c1_0_w(op, f[0], c0, c1)
    v1 = op.mul3(f[1][1], c4)  # f.c1 * (0, c4, 0)
    t0 = op.mul3(f[1][1] + f[1][2], c4)
    t1 = op.mul3(f[1][0] + f[1][1], c4)
    bb = [(t0 - v1).xi(), t1 - v1, v1]
    t = fp6_mul_by_c0_c1_0_w(op, fp6_add(f[0], f[1]), c0, c1 + c4)
    return [fp6_add(fp6_mul_v(bb), aa), fp6_sub(fp6_sub(t, aa), bb)]


def reg(base):
    s = [LC.slot(base + j) for j in range(6)]
    return [s[0:3], s[3:6]]


def finish(op, res):
    op.out = res[0] + res[1]
    return op


def build_ops():
    ops = []
    a, b = reg(IN0), reg(IN1)

    op = Op("MUL")  # fp12_mul_w
    v0 = fp6_mul_w(op, a[0], b[0])
    v1 = fp6_mul_w(op, a[1], b[1])
    s = fp6_mul_w(op, fp6_add(a[1], a[0]), fp6_add(b[0], b[1]))
    ops.append(finish(op, [fp6_add(v0, fp6_mul_v(v1)), fp6_sub(fp6_sub(s, v0), v1)]))

    op = Op("SQR")  # fp12_sqr_w
    w0 = fp6_sub(a[0], a[1])
    w3 = fp6_sub(a[0], fp6_mul_v(a[1]))
    v2 = fp6_mul_w(op, a[0], a[1])
    t = fp6_add(fp6_mul_w(op, w0, w3), v2)
    ops.append(finish(op, [fp6_add(t, fp6_mul_v(v2)), [x.dbl() for x in v2]]))

    op = Op("CYC")  # fp12_cyclotomic_square_w
    z0, z4, z3, z2, z1, z5 = a[0][0], a[0][1], a[0][2], a[1][0], a[1][1], a[1][2]

    def half(za, zb):
        tmp = op.mul3(za, zb)
        prod = op.mul3(za + zb, zb.xi() + za)
        return prod - (tmp.xi() + tmp), tmp.dbl()

    t0, t1 = half(z0, z1)
    t2, t3 = half(z2, z3)
    t4, t5 = half(z4, z5)
    c0_c0 = (t0 - z0).dbl() + t0
    c1_c1 = (t1 + z1).dbl() + t1
    xt5 = t5.xi()
    c1_c0 = (z2 + xt5).dbl() + xt5
    c0_c2 = (t4 - z3).dbl() + t4
    c0_c1 = (t2 - z4).dbl() + t2
    c1_c2 = (t3 + z5).dbl() + t3
    ops.append(finish(op, [[c0_c0, c0_c1, c0_c2], [c1_c0, c1_c1, c1_c2]]))

    op = Op("ELLC")  # ell for (-g1 constant, sig): c1 already multiplied by g1.x, y constant
    ops.append(finish(op, fp12_mul_by_014_w(op, a, LC.slot(XS0), LC.slot(XS1), LC.slot(XYC), False)))

    op = Op("ELLV")  # ell for (pk, H(m)): k0 = c1.c0*px, k1 = c1.c1*px are witnesses, y = pk.y variable
    c1 = op.mul2b(LC.slot(XH1), LC.slot(XPX))
    ops.append(finish(op, fp12_mul_by_014_w(op, a, LC.slot(XH0), c1, LC.slot(XYV), True)))

    # ---- native (value-only) pairing of blsw_verify_batch: ell with projective line coefficients (c0, c1 * p.x, c2 * p.y) in the pair slots
    op = Op("ELLGS", value_only=True)  # the (-g1, sig) pair: XS0, XS1, XYC
    ops.append(finish(op, fp12_mul_by_014_general(op, a, LC.slot(XS0), LC.slot(XS1), LC.slot(XYC))))
    op = Op("ELLGH", value_only=True)  # the (pk, H(m)) pair: XH0, XH1, XYV
    ops.append(finish(op, fp12_mul_by_014_general(op, a, LC.slot(XH0), LC.slot(XH1), LC.slot(XYV))))

    # ---- G2 points, homogeneous projective (x, y, z) on lanes 0..2 (curve.hpp: proj_double_w / proj_add_w<0> over Fp2;
    # 3b = 12 xi). IN0 = p (slots 0..2), IN1 = q (slots 6..8).
    px, py, pz = LC.slot(IN0 + 0), LC.slot(IN0 + 1), LC.slot(IN0 + 2)
    qx, qy, qz = LC.slot(IN1 + 0), LC.slot(IN1 + 1), LC.slot(IN1 + 2)

    def x12xi(v):
        t = v.xi()
        return LC({k: (a, 12 * b) for k, (a, b) in t.d.items()})

    op = Op("G2DBL")
    xx = op.sqr2(px)
    yy = op.sqr2(py)
    zz = op.sqr2(pz)
    xy2 = op.mul3(px, py).dbl()
    xz2 = op.mul3(px, pz).dbl()
    bzz3 = x12xi(zz)
    yy_m, yy_p = yy - bzz3, yy + bzz3
    y_frag = op.mul3(yy_p, yy_m)
    x_frag = op.mul3(yy_m, xy2)
    bxz3 = x12xi(xz2)
    xx3 = xx.dbl() + xx
    tt = op.mul3(xx3, bxz3)
    yz2 = op.mul3(py, pz).dbl()
    t2 = op.mul3(bxz3, yz2)
    zp = op.mul3(yz2, yy)
    op.out = [x_frag - t2, y_frag + tt, zp.dbl().dbl(), ZERO, ZERO, ZERO]
    ops.append(op)

    op = Op("G2ADD")
    xx = op.mul3(px, qx)
    yy = op.mul3(py, qy)
    zz = op.mul3(pz, qz)
    xy_pairs = op.mul3(px + py, qx + qy) - (xx + yy)
    xz_pairs = op.mul3(px + pz, qx + qz) - (xx + zz)
    yz_pairs = op.mul3(py + pz, qy + qz) - (yy + zz)
    bzz3 = x12xi(zz)
    yy_m, yy_p = yy - bzz3, yy + bzz3
    xx3 = xx.dbl() + xx
    bxz3 = x12xi(xz_pairs)
    m0 = op.mul3(yy_m, xy_pairs)
    m1 = op.mul3(yz_pairs, bxz3)
    m2 = op.mul3(yy_p, yy_m)
    m3 = op.mul3(xx3, bxz3)
    m4 = op.mul3(yy_p, yz_pairs)
    m5 = op.mul3(xy_pairs, xx3)
    op.out = [m0 - m1, m2 + m3, m4 + m5, ZERO, ZERO, ZERO]
    ops.append(op)

    op = Op("INVCHK")  # mul_equals(self, inverse, one) of fp12_inv_w: IN0 = self, IN1 = inverse
    fp6_mul_w(op, a[1], b[1])
    fp6_mul_equals_w(op, a[0], b[0])
    fp6_mul_equals_w(op, fp6_add(a[0], a[1]), fp6_add(b[0], b[1]))
    op.out = [ZERO] * 6
    ops.append(op)
    return ops


def c_lin(lc):
    lp, ln, mp, mn = lc.lists()
    idx = lp + ln + mp + mn
    idx += [0] * (MAX_TERMS - len(idx))
    n = len(lp) | len(ln) << 8 | len(mp) << 16 | len(mn) << 24 | {1: 0, 12: 1, 24: 2}[lc.mscale()] << 6
    words = [idx[4 * w] | idx[4 * w + 1] << 8 | idx[4 * w + 2] << 16 | idx[4 * w + 3] << 24 for w in range(4)]
    return "{0x%08xu, {%s}}" % (n, ", ".join("0x%08xu" % w for w in words))


def c_task(kind, dst, woff, a, b):
    return "{0x%08xu, %s, %s}" % (kind | dst << 8 | woff << 16, c_lin(a), c_lin(b))


def main():
    ops = build_ops()
    out = []
    w = out.append
    w("// GENERATED by tools/gen_team_tables.py — do not edit. Op tables of the 6-lanes-per-instance pairing kernel (team.hpp).")
    w("#pragma once")
    w("#include <stdint.h>")
    w("namespace blsw {")
    w("enum { %s };" % ", ".join("%s = %d" % (n, i) for i, n in enumerate(KIND_NAMES)))
    w("enum { TS_IN0 = %d, TS_IN1 = %d, TS_P = %d, TS_XS0 = %d, TS_XS1 = %d, TS_XH0 = %d, TS_XH1 = %d, TS_XYC = %d, TS_XYV = %d, TS_XPX = %d, TS_NSLOTS = %d };"
      % (IN0, IN1, P0, XS0, XS1, XH0, XH1, XYC, XYV, XPX, N_SLOTS))
    w("enum { TEAM_MAX_TERMS = %d, TEAM_MAX_ROUNDS = 4 };" % MAX_TERMS)
    w("// (sum Lp - sum Ln) + xi * (sum Mp - sum Mn): n = the four counts, one byte each; idx = 16 slot numbers, one byte each.")
    w("// Whole descriptors are loaded into registers with a few wide loads (never indexed in memory on the device).")
    w("struct TeamLin { uint32_t n; uint32_t idx[4]; };")
    w("struct TeamTask { uint32_t hdr; TeamLin a, b; };  // hdr = kind | result slot << 8 | witness offset << 16")
    w("struct TeamOp { uint32_t rounds, n_witness; TeamTask task[TEAM_MAX_ROUNDS][6]; TeamLin out[6]; };")
    w("#if defined(__HIPCC__)")
    w("#define BLSW_TEAM_TABLE __constant__ const")
    w("#else")
    w("#define BLSW_TEAM_TABLE static const")
    w("#endif")
    none_task = c_task(K_NONE, 0xFF, 0, ZERO, ZERO)
    summary = []
    for op in ops:
        rounds = op.schedule()
        assert len(rounds) <= 4, (op.name, len(rounds))
        w("BLSW_TEAM_TABLE TeamOp TEAM_OP_%s = {%d, %d, {" % (op.name, len(rounds), op.woff))
        for r in range(4):
            row = []
            for j in range(6):
                if r < len(rounds) and j < len(rounds[r]):
                    t = op.tasks[rounds[r][j]]
                    row.append(c_task(t["kind"], t["dst"], t["woff"], t["a"], t["b"]))
                else:
                    row.append(none_task)
            w("    {" + ",\n     ".join(row) + "},")
        w("  }, {" + ",\n      ".join(c_lin(o) for o in op.out) + "}};")
        summary.append("%s: %d tasks in %d rounds %s, %d witnesses, longest output %d terms" % (
            op.name, len(op.tasks), len(rounds), [len(r) for r in rounds], op.woff, max(sum(map(len, o.lists())) for o in op.out)))
    w("}  // namespace blsw")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bls-verify-gadget_amd", "csrc", "team_tables.hpp")
    if len(sys.argv) > 1:
        path = sys.argv[1]
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(summary), file=sys.stderr)


if __name__ == "__main__":
    main()
