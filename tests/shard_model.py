"""Executable model (big ints) of the multi-GPU decomposition in prover.hip / poly.hip — the algebra the
HIP kernels implement, checked against the oracle on the CPU (tests/test_sharding_classes_gloo.py):

  * the quotient coset {g w_N^i}, N = Q n, as Q size-n cosets ("classes") j = i mod Q with shift g w_N^j;
  * a polynomial with more than n coefficients folded mod (X^n - sigma_j), sigma_j = (g w_N^j)^n (fold_kernel);
  * per class: evaluations -> size-n inverse coset transform -> remainder F_j = t mod (X^n - sigma_j);
  * the exchange (shard_pack_kernel / all-to-all): rank r receives F_j[lo_r .. lo_r + per) and F_j[0..8) of every class;
  * shard_combine_kernel: Q-point inverse DFT with coef[i1][j] = g^(-n i1) w_Q^(-j i1) / Q, de-aliasing for Q = 4;
  * range-sharded evaluation (sum_r x^lo_r * partial_r) and ruffini (suffix sums + carry of the ranks above).
"""
from oracle.bls12_381 import GENERATOR, Q as MOD, ROOT_OF_UNITY
from oracle.fft import EvaluationDomain


def omega(log_size):
    return pow(ROOT_OF_UNITY, 1 << (32 - log_size), MOD)


def coset_ntt(coeffs, n, shift):
    """evaluations of a polynomial with <= n coefficients at shift * w_n^k"""
    d = EvaluationDomain(n)
    scaled, p = [], 1
    for c in coeffs:
        scaled.append(c * p % MOD)
        p = p * shift % MOD
    return d.fft(scaled)


def coset_intt(evals, n, shift):
    d = EvaluationDomain(n)
    c = d.ifft(evals)
    inv, p, out = pow(shift, -1, MOD), 1, []
    for v in c:
        out.append(v * p % MOD)
        p = p * inv % MOD
    return out


class Layout:
    def __init__(self, n, world, srs_total=None):
        self.n, self.W = n, world
        self.Q = 8 if world == 8 else 4
        self.cpr = self.Q // world
        self.logN = (self.Q * n).bit_length() - 1
        self.wN = omega(self.logN)
        self.srs_total = srs_total if srs_total is not None else n + 7
        self.per = -(-self.srs_total // world)
        self.gn = pow(GENERATOR, n, MOD)
        self.wQ = pow(self.wN, n, MOD)                      # primitive Q-th root

    def classes(self, rank):
        return [rank + self.W * k for k in range(self.cpr)]

    def shift(self, j):
        return GENERATOR * pow(self.wN, j, MOD) % MOD

    def sigma(self, j):                                     # x^n on class j
        return self.gn * pow(self.wQ, j, MOD) % MOD

    def rng(self, rank):
        lo = min(self.per * rank, self.n + 7)
        return lo, min(self.per * rank + self.per, self.n + 7)

    def coef(self, i1, j):
        return pow(self.gn, -i1, MOD) * pow(self.wQ, -(j * i1), MOD) % MOD * pow(self.Q, -1, MOD) % MOD


def fold(poly, n, sigma):
    out = [c % MOD for c in poly[:n]] + [0] * max(0, n - len(poly))
    for i, c in enumerate(poly[n:]):
        out[i] = (out[i] + sigma * c) % MOD
    return out


def class_evals(lay, poly, j):
    """a polynomial (<= n + 8 coefficients) on class j: fold, then a size-n coset transform"""
    return coset_ntt(fold(poly, lay.n, lay.sigma(j)), lay.n, lay.shift(j))


def remainder(lay, t_evals_on_class, j):
    return coset_intt(t_evals_on_class, lay.n, lay.shift(j))


def pack(lay, F_by_class, rank):
    """send[peer][k] = F_k[peer * per : +per] (zero beyond n) ++ F_k[0:8]"""
    msgs = []
    for peer in range(lay.W):
        row = []
        for j in lay.classes(rank):
            F = F_by_class[j]
            body = [F[i] if i < lay.n else 0 for i in range(peer * lay.per, peer * lay.per + lay.per)]
            row.append(body + F[:8])
        msgs.append(row)
    return msgs


def combine(lay, recv, rank, low7=None):
    """recv[src][k] = message of class src + W k.  Returns {(part, index): value} for the owned range."""
    lo, hi = lay.rng(rank)
    msg = {}
    for src in range(lay.W):
        for k in range(lay.cpr):
            msg[src + lay.W * k] = recv[src][k]
    out = {}

    def dft(i1, off):
        return sum(lay.coef(i1, j) * msg[j][off] for j in range(lay.Q)) % MOD
    for idx in range(lo, min(hi, lay.n)):
        for i1 in range(4):
            if i1 == 0 and lay.Q == 4 and idx < 7:
                continue
            out[(i1, idx)] = dft(i1, idx - lo)
    g4n_inv = pow(GENERATOR, -(4 * lay.n), MOD)
    for k in range(7):
        if lay.Q == 4:
            top = (dft(0, lay.per + k) - low7[k]) * g4n_inv % MOD
            if lo <= k < hi:
                out[(0, k)] = low7[k] % MOD
        else:
            top = dft(4, lay.per + k)
        if lo <= lay.n + k < hi:
            out[(3, lay.n + k)] = top
    return out


def eval_partial(poly, lo, hi, x):
    return sum(c * pow(x, i, MOD) for i, c in enumerate(poly[lo:hi])) % MOD


def ruffini_local(poly, lo, hi, z):
    """suffix sums S[i] = sum_{j >= i, j < hi} c_j z^j for i in [lo, hi], S[hi] = 0; S[lo] is the range total"""
    S, acc = [0] * (hi - lo + 1), 0
    for i in range(hi - 1, lo - 1, -1):
        acc = (acc + poly[i] * pow(z, i, MOD)) % MOD
        S[i - lo] = acc
    return S


def ruffini_finish(S, lo, hi, z, carry):
    zi = pow(z, -1, MOD)
    return {lo + i: (S[i + 1] + carry) * pow(zi, lo + i + 1, MOD) % MOD for i in range(hi - lo)}


# ---- round 2, sharded (round 4): the permutation grand product by evaluation-index range --------------------------------
def grand_product_terms(n, wires, sigma_ev, beta, gamma, first, count):
    """perm_terms on [first, first + count): t[0] = 1, t[i] = prod_j (w_j[i-1] + beta K_j w^(i-1) + gamma) / (w_j[i-1] + beta sigma_j[i-1] + gamma)
    (permutation.rs:213-294 with the reference's shift: z[i] is the product of the ratios at rows < i)"""
    ks = (1, 7, 13, 17)
    w = omega(n.bit_length() - 1)
    out = []
    for i in range(first, first + count):
        if i == 0:
            out.append(1)
            continue
        r = i - 1
        root = pow(w, r, MOD)
        num = den = 1
        for j in range(4):
            num = num * ((wires[j][r] + beta * ks[j] % MOD * root + gamma) % MOD) % MOD
            den = den * ((wires[j][r] + beta * sigma_ev[j][r] + gamma) % MOD) % MOD
        out.append(num * pow(den, -1, MOD) % MOD)
    return out


def grand_product_rank(terms):
    """local inclusive prefix products of a rank's terms and the range product (prover.hip: scan_prefix_product_local)"""
    acc, out = 1, []
    for t in terms:
        acc = acc * t % MOD
        out.append(acc)
    return out, acc


def grand_product_finish(local, totals, rank):
    """scale by the product of the ranges before this one (scan_prefix_product_apply): the rank's slice of z's evaluations"""
    carry = 1
    for t in totals[:rank]:
        carry = carry * t % MOD
    return [v * carry % MOD for v in local]
