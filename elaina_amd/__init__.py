"""elaina_amd -- MI355X-native Walk-on-Stars hot path behind Elaina's integrator surface.

The product is libwost_hip.so (hand-written HIP for gfx950, C-ABI in include/wost.h) plus
the C++ host mirror of the reference's exec.h / problem.h / integrator classes in
elaina_amd/host/.  The Python modules here only bind the C-ABI for tests and bench.py.
"""
from .problem import Problem  # noqa: F401
from .integrator import UniformIntegrator, UniformIntegratorSettings  # noqa: F401
