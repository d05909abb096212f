"""EvaluationDomain restatement (oracle; test infrastructure only).

Follows reference src/fft/domain.rs line by line in structure:
  new                      domain.rs:122-158
  fft / fft_in_place       domain.rs:166-176   (resize => zero-pad OR TRUNCATE, :174)
  ifft / ifft_in_place     domain.rs:179-196
  distribute_powers        domain.rs:198-204
  coset_fft / coset_ifft   domain.rs:207-232
  serial_fft               domain.rs:443-463   (bit-reverse + log n DIT stages,
                                                twiddles by running product :480-488)
Values are canonical ints mod Q (the Montgomery form is a storage detail).
"""
from __future__ import annotations

from .bls12_381 import GENERATOR, Q, ROOT_OF_UNITY, TWO_ADACITY, fr_inv


def next_pow2(n: int) -> int:
    return 1 if n <= 1 else 1 << (n - 1).bit_length()


def bitreverse(n: int, l: int) -> int:            # domain.rs:425-432
    r = 0
    for _ in range(l):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def serial_fft(a: list[int], omega: int, log_n: int) -> None:   # domain.rs:443-463
    n = len(a)
    assert n == 1 << log_n
    for k in range(n):                             # bitreverse_permute :434-441
        rk = bitreverse(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), Q)
        for start in range(0, n, 2 * m):           # butterfly_chunk :466-489
            w = 1
            for j in range(m):
                t = a[start + m + j] * w % Q
                left = a[start + j]
                a[start + m + j] = (left - t) % Q
                a[start + j] = (left + t) % Q
                w = w * w_m % Q
        m *= 2


class EvaluationDomain:
    def __init__(self, num_coeffs: int):           # domain.rs:122-158
        self.size = next_pow2(num_coeffs)
        self.log_size_of_group = self.size.bit_length() - 1
        if self.log_size_of_group >= TWO_ADACITY:
            raise ValueError("InvalidEvalDomainSize")
        g = ROOT_OF_UNITY
        for _ in range(self.log_size_of_group, TWO_ADACITY):
            g = g * g % Q
        self.group_gen = g
        self.group_gen_inv = fr_inv(g)
        self.size_as_field_element = self.size % Q
        self.size_inv = fr_inv(self.size)
        self.generator_inv = fr_inv(GENERATOR)

    def _resize(self, v: list[int]) -> list[int]:  # Vec::resize at :174,:188
        v = list(v[: self.size])
        return v + [0] * (self.size - len(v))

    def fft(self, coeffs):                         # :166-176
        a = self._resize(coeffs)
        serial_fft(a, self.group_gen, self.log_size_of_group)
        return a

    def ifft(self, evals):                         # :179-196
        a = self._resize(evals)
        serial_fft(a, self.group_gen_inv, self.log_size_of_group)
        return [x * self.size_inv % Q for x in a]

    @staticmethod
    def distribute_powers(coeffs, g):              # :198-204
        out, p = [], 1
        for c in coeffs:
            out.append(c * p % Q)
            p = p * g % Q
        return out

    def coset_fft(self, coeffs):                   # :207-218 (scale BEFORE resize)
        return self.fft(self.distribute_powers(coeffs, GENERATOR))

    def coset_ifft(self, evals):                   # :221-232
        return self.distribute_powers(self.ifft(evals), self.generator_inv)

    def elements(self):                            # :354-360,526-538
        out, cur = [], 1
        for _ in range(self.size):
            out.append(cur)
            cur = cur * self.group_gen % Q
        return out

    def evaluate_vanishing_polynomial(self, tau):  # :289-294
        return (pow(tau, self.size, Q) - 1) % Q

    def vanishing_poly_over_coset(self, poly_degree: int):   # :338-351
        assert self.size > poly_degree
        point = pow(GENERATOR, poly_degree, Q)
        step = pow(self.group_gen, poly_degree, Q)
        out = []
        for _ in range(self.size):
            out.append((point - 1) % Q)
            point = point * step % Q
        return out

    def first_lagrange_at(self, tau):
        """evaluate_all_lagrange_coefficients(tau)[0]  (:237-284)."""
        t_size = pow(tau, self.size, Q)
        if t_size == 1:
            return 1 if tau == 1 else 0
        l = (t_size - 1) * self.size_inv % Q
        return l * fr_inv((tau - 1) % Q) % Q
