"""Results one GPU test hands to later tests of the same session, so that the 2^20 / 2^22 single-GPU provers are built once:
SINGLE[(log_gates, profile)] = the fields of a single-GPU bench line the multi-rank tests compare against (proof digest).
Filled by tests/test_gpu_fullsize.py (which checks those very proofs against the verification equation and, at 2^22, against
the C oracle byte for byte); tests/test_gpu_multirank.py::single falls back to a bench.py child when an entry is missing."""
SINGLE = {}
