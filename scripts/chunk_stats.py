"""Prints the chunk statistics of a configuration (verbose build log of the handle)."""
import sys
sys.path.insert(0, '.')
from mpsfm_amd import capi
from mpsfm_amd.synthetic import make_config
prob, _ = make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
opts = capi.default_options()
opts.verbose = 2
h = capi.BAHandle(prob, opts)
