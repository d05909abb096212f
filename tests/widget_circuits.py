"""Circuits whose custom gates carry NON-TRIVIAL, semantically constructed witnesses.

Each gadget below lays out its rows exactly as the reference composer does and fills them from
the *meaning* of the gate (bit quads, XOR / AND truth, twisted-Edwards addition on JubJub), not
from the constraint formulas.  A wrong restatement of a widget identity (oracle, HIP quotient
kernel, or the F[X]/(X^7) host evaluation of the 4n path) would make these honest witnesses
"unsatisfied" and fail the tests, so they pin the formulas to the gates' semantics:

  range        composer/range.rs:68-130 + widget/range/proverkey.rs:32-58
  logic        composer/logic.rs:42-170 (accumulator table in the comment at :76-88), q_c = q_logic
               = +1 (AND) / -1 (XOR) per constraint_system/constraint.rs:211-221
  fixed base   composer/fixed_base.rs:160-290: rows (acc_x, acc_y, xy_alpha, scalar_acc) with
               q_l = x_beta, q_r = y_beta, q_c = xy_beta, closing anchor row
  curve add    composer/point.rs:356-408: (x1, y1, x2, y2) then (x3, y3, 0, x1*y2)
"""
import random

from oracle import plonk as O
from oracle.bls12_381 import Q

EDWARDS_D = (-10240 * pow(10241, -1, Q)) % Q       # dusk_jubjub::EDWARDS_D, curve -x^2 + y^2 = 1 + d x^2 y^2


def fr_sqrt(a):
    """Tonelli-Shanks in Fr (2-adicity 32); None for a non-residue."""
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        return None
    s, t = 32, (Q - 1) >> 32
    z = pow(7, t, Q)                    # 7 generates Fr*
    m, c, r, b = s, z, pow(a, (t + 1) // 2, Q), pow(a, t, Q)
    while b != 1:
        i, x = 0, b
        while x != 1:
            x = x * x % Q
            i += 1
        f = pow(c, 1 << (m - i - 1), Q)
        m, c, r, b = i, f * f % Q, r * f % Q, b * f * f % Q
    return r


def on_curve(p):
    x, y = p
    return (-x * x + y * y - 1 - EDWARDS_D * x * x % Q * y * y) % Q == 0


def jj_add(p1, p2):
    (x1, y1), (x2, y2) = p1, p2
    k = EDWARDS_D * x1 % Q * x2 % Q * y1 % Q * y2 % Q
    return ((x1 * y2 + y1 * x2) * pow(1 + k, -1, Q) % Q, (y1 * y2 + x1 * x2) * pow(1 - k, -1, Q) % Q)


def jj_neg(p):
    return ((-p[0]) % Q, p[1])


def jj_base():
    """Some point of the curve (the gates do not care about subgroup membership)."""
    y = 2
    while True:
        x = fr_sqrt((y * y - 1) * pow(EDWARDS_D * y * y + 1, -1, Q))
        if x:
            assert on_curve((x, y))
            return (x, y)
        y += 1


def add_range(c, value, quads):
    """component_range: accumulators of 2-bit quads, 4 per row (d, c, b, a), next row's d continues."""
    digits = [(value >> (2 * (quads - 1 - i))) & 3 for i in range(quads)]   # most significant first
    accs, acc = [0], 0
    for q in digits:
        acc = 4 * acc + q
        accs.append(acc)
    assert len(accs) % 4 == 1                       # rows of (a, b, c, d) with d shared as next row's start
    wit = [c.append_witness(v) for v in accs]
    rows = (len(accs) - 1) // 4
    for r in range(rows):
        d, cc, b, a = wit[4 * r], wit[4 * r + 1], wit[4 * r + 2], wit[4 * r + 3]
        c.append_custom_gate(O.Gate(a=a, b=b, c=cc, d=d, q_range=1))
    c.append_gate(O.Gate(d=wit[-1]))                # row carrying the final accumulator as d_next
    return wit[-1]


def add_logic(c, x, y, quads, xor):
    """append_logic_component: rows (a_i, b_i, w_{i+1}, d_i), i = 0..quads-1, then (a_n, b_n, 0, d_n)."""
    sel = Q - 1 if xor else 1
    la = [(x >> (2 * (quads - 1 - i))) & 3 for i in range(quads)]
    lb = [(y >> (2 * (quads - 1 - i))) & 3 for i in range(quads)]
    a = b = d = 0
    wa = wb = wd = 0                                # witness 0 is ZERO
    for i in range(quads):
        wc = c.append_witness(la[i] * lb[i])
        c.append_custom_gate(O.Gate(a=wa, b=wb, c=wc, d=wd, q_c=sel, q_logic=sel))
        a, b = 4 * a + la[i], 4 * b + lb[i]
        d = 4 * d + ((la[i] ^ lb[i]) if xor else (la[i] & lb[i]))
        wa, wb, wd = c.append_witness(a), c.append_witness(b), c.append_witness(d)
    c.append_custom_gate(O.Gate(a=wa, b=wb, d=wd))
    assert d == ((x ^ y) if xor else (x & y)) & ((1 << (2 * quads)) - 1)
    return wd


def add_fixed_base(c, base, digits):
    """append_fixed_base_signed_digits with len(digits) rounds of signed digits in {-1, 0, 1}."""
    rounds = len(digits)
    mult = [base]
    for _ in range(1, rounds):
        mult.append(jj_add(mult[-1], mult[-1]))
    mult.reverse()                                   # most significant first
    acc_pt, acc_sc = (0, 1), 0                       # identity, zero
    for i, dg in enumerate(digits):
        pt = (0, 1) if dg == 0 else (mult[i] if dg == 1 else jj_neg(mult[i]))
        xb, yb = mult[i]
        wx, wy, ws = c.append_witness(acc_pt[0]), c.append_witness(acc_pt[1]), c.append_witness(acc_sc)
        wxy = c.append_witness(pt[0] * pt[1])
        c.append_custom_gate(O.Gate(a=wx, b=wy, c=wxy, d=ws, q_l=xb, q_r=yb, q_c=xb * yb % Q, q_fixed_group_add=1))
        acc_pt, acc_sc = jj_add(acc_pt, pt), (2 * acc_sc + dg) % Q
    c.append_gate(O.Gate(a=c.append_witness(acc_pt[0]), b=c.append_witness(acc_pt[1]), d=c.append_witness(acc_sc)))
    assert on_curve(acc_pt)
    return acc_pt


def add_curve_addition(c, p1, p2):
    """add_point_gates: (x1, y1, x2, y2) with q_variable_group_add, then (x3, y3, 0, x1 * y2)."""
    p3 = jj_add(p1, p2)
    w = [c.append_witness(v) for v in (*p1, *p2)]
    c.append_custom_gate(O.Gate(a=w[0], b=w[1], c=w[2], d=w[3], q_variable_group_add=1))
    c.append_custom_gate(O.Gate(a=c.append_witness(p3[0]), b=c.append_witness(p3[1]), d=c.append_witness(p1[0] * p2[1])))
    return p3


def semantic_widget_circuit(seed=1):
    """Every widget family with non-trivial honest witnesses, arithmetic gates and two public inputs."""
    def build():
        r = random.Random(seed)
        c = O.Composer()
        add_range(c, r.getrandbits(32), 16)
        add_logic(c, r.getrandbits(20), r.getrandbits(20), 10, xor=True)
        add_logic(c, r.getrandbits(20), r.getrandbits(20), 10, xor=False)
        base = jj_base()
        p = add_fixed_base(c, base, [0, 0] + [r.choice((-1, 0, 1)) for _ in range(14)])
        q = add_fixed_base(c, base, [1] + [r.choice((-1, 0, 1)) for _ in range(9)])
        s = add_curve_addition(c, p, q)
        add_curve_addition(c, s, s)                  # doubling through the unified addition law
        ws = [c.append_witness(r.randrange(Q)) for _ in range(4)]
        for _ in range(6):
            ws.append(c.gate_mul(r.choice(ws), r.choice(ws), r.choice(ws), q_m=r.randrange(1, Q), q_f=1, q_c=r.randrange(Q)))
        for _ in range(2):
            v = r.randrange(Q)
            c.append_gate(O.Gate(a=c.append_witness(v), q_l=Q - 1, pi=v))   # append_public
        return c
    return build
