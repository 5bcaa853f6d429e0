"""CPU suite: the drop-in command line's readers (cnf2freq_amd/csrc/host), exercised through
`cnF2freq --parse-only` (no GPU needed).  Expected values follow readalphamap / readalphaped /
readalphadata (cnF2freq.cpp:6495-6685) on hand-written inputs in the demo's format (CRLF)."""
import os
import subprocess

import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "cnf2freq_amd", "cnF2freq")


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(EXE)
    return EXE


def write_inputs(d):
    (d / "t.map").write_bytes(b"0\r\n10\r\n20\r\n30\r\n0\r\n5\r\n6\r\n")
    (d / "t.ped").write_bytes(b"A 0 0\r\nB 0 0\r\nC A B 2\r\nD A B 2\r\nE A B 1\r\nF E H 20\r\n")
    (d / "t.gen").write_bytes(b"A 2 2 2 0 0 0 2\r\nB 2 0 2 1 2 1/0 2\r\nC 9 1 9 9 0 1 9\r\nD 2 1 2 5/1 1 1/0 1\r\n")


def parse(exe, d, extra=()):
    out = subprocess.run([exe, "--mapfile", str(d / "t.map"), "--pedfile", str(d / "t.ped"), "--genfile",
                          str(d / "t.gen"), "--parse-only", "--quiet", *extra], capture_output=True, text=True, check=True)
    lines = out.stdout.strip().splitlines()
    inds = {}
    for ln in lines[2:]:
        head, geno = ln.split(" :")
        f = head.split()
        inds[f[2]] = dict(n=int(f[1]), gen=int(f[4]), empty=int(f[6]), pars=(int(f[8]), int(f[9])), row=int(f[11]),
                          analysed=int(f[13]), geno=geno.split())
    return lines[0], lines[1], inds


def test_map_ped_gen(exe, tmp_path):
    write_inputs(tmp_path)
    head, rows, inds = parse(exe, tmp_path)
    # a position smaller than its predecessor starts a new chromosome (cnF2freq.cpp:6676-6679)
    assert head == "markers 7 chromstarts 0 4 7"
    # numbering in order of first mention; "0" is nobody (cnF2freq.cpp:6480-6493)
    assert [inds[k]["n"] for k in ("A", "B", "C", "C_aux_realf", "C_aux_realm", "D")] == [1, 2, 3, 4, 5, 6]
    # gen >= 2 with two generation-0 parents gets private empty F1 parents (cnF2freq.cpp:6515-6527)
    assert inds["C"]["pars"] == (4, 5) and inds["C_aux_realf"]["pars"] == (1, 2) and inds["C_aux_realf"]["gen"] == 1
    assert inds["C_aux_realf"]["empty"] == 1 and inds["C_aux_realf"]["row"] == 0
    # a listed generation-1 parent is used directly (cnF2freq.cpp:6528-6533); H appears only as a parent
    assert inds["F"]["pars"] == (inds["E"]["n"], inds["H"]["n"]) and inds["F"]["gen"] == 20
    assert [k for k in inds if inds[k]["analysed"]] == ["C", "D", "F"]
    # genotype tokens 0/1/2 -> (1,1)/(1,2)/(2,2) with sure 0.02, anything else unknown (cnF2freq.cpp:6568-6587)
    assert inds["A"]["geno"][0] == "22/0.02/0.02" and inds["A"]["geno"][3] == "11/0.02/0.02"
    assert inds["C"]["geno"][0] == "00/0/0" and inds["C"]["geno"][1] == "12/0.02/0.02"
    # read counts a/b -> binomial-posterior error rates (cnF2freq.cpp:6589-6657)
    assert inds["B"]["geno"][5] == "11/0/0.5"
    a, s1, s2 = inds["D"]["geno"][3].split("/")
    assert a == "11" and abs(float(s1) - 0.00212119) < 1e-7 and abs(float(s2) - 0.47464) < 1e-5
    # individuals with a genotype line are no longer empty; the "haplo" pseudo-individual exists
    assert inds["C"]["empty"] == 0 and inds["E"]["empty"] == 1 and inds["haplo"]["geno"][0] == "99/0/0"
    assert rows == "rows 6"


def test_capmarker(exe, tmp_path):
    write_inputs(tmp_path)
    head, _, inds = parse(exe, tmp_path, ("--capmarker", "5"))
    assert head == "markers 5 chromstarts 0 4 5"
    assert len(inds["A"]["geno"]) == 5
