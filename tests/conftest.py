import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
GOLDEN_CASES = ["f2_implicit_f1", "outbred3_missing", "random_windows", "f2_ungenotyped"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """(Pedigree rebuilt from the fixture's inputs, dict of reference outputs)."""
    from cnf2freq_amd.synth import Pedigree
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    R = len(z["in_par"])
    ped = Pedigree(["r%d" % i for i in range(R)], z["in_par"], z["in_gen"], z["in_empty"], z["in_row_of"],
                   z["in_allele"], z["in_sure"], z["in_hw"], z["in_pos"], z["in_chromstarts"], z["in_dous"])
    ped.founder_flags()
    return ped, z


def oracle_ped(ped):
    from oracle.pyoracle import OraclePed
    a, s, h = ped.dense()
    return OraclePed(a, s, h, ped.par, ped.empty, ped.pos)


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return load_golden(request.param)
