"""Test-only solver backends for the Optimizer shim (the product backend is mpsfm_amd...HipBackend)."""
from oracle import cpu_oracle as O
from oracle import prior_oracle


class OracleBackend:
    """Runs the assembled flat problem on the CPU oracle and the depth-block arithmetic on its NumPy restatement."""

    def solve(self, prob):
        return O.solve(prob)

    def point_covs(self, prob):
        return O.point_covs(prob)

    def depth_blocks(self, **kw):
        return prior_oracle.depth_blocks(**kw)
