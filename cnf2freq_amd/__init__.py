"""cnf2freq_amd -- MI355X-native forward-backward sweep of cnF2freq.

csrc/     HIP kernels (gfx950) + the C ABI of include/cnf2hip.h (libcnf2hip.so),
          csrc/host: C++ readers and the `cnF2freq` drop-in command line
capi.py   ctypes mirror of the C ABI for tests and bench.py
synth.py  deterministic synthetic pedigrees
"""
from . import synth  # noqa: F401
