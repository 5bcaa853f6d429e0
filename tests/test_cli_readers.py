"""CPU suite: the drop-in command line's readers (cnf2freq_amd/csrc/host), exercised through
`cnF2freq --parse-only` (no GPU needed).  Expected values follow readalphamap / readalphaped /
readalphadata (cnF2freq.cpp:6495-6685) on hand-written inputs in the demo's format (CRLF)."""
import os
import subprocess

import pytest

from conftest import ROOT

# CNF2_EXE: another build of the executable (tools/sanitize_host.sh runs these tests on one made with -fsanitize=address,undefined)
EXE = os.environ.get("CNF2_EXE") or os.path.join(ROOT, "cnf2freq_amd", "cnF2freq")


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


def test_genotype_tokens_across_the_read_buffer(exe, tmp_path):
    """The genotype reader takes its tokens from a 4 MB buffer (cnf2_readers.cpp TokenReader) and must deliver what
    fscanf("%254s") + sscanf("%d/%d") deliver (cnF2freq.cpp:6551-6563): any run of blanks, tabs, CR and LF separates tokens,
    a token that is not a number leaves `data` at the previous token's value (it is declared outside the marker loop) and goes
    down the read-count branch, a file that ends inside a line leaves the rest of the line to that branch too.  8 MB of
    tokens of every kind, with separators of random width, so that tokens of several characters straddle the buffer's end."""
    import random
    rng = random.Random(5)
    M, N = 3000, 700
    (tmp_path / "t.map").write_text("\n".join(str(i) for i in range(M)) + "\n")
    (tmp_path / "t.ped").write_text("")
    kinds = ["0", "1", "2", "9", "5/1", "1/0", "12", "0/0", "007"]
    seps = [" ", "  ", "\t", " \t ", "\r\n ", "\n"]
    expect = {"0": "11/0.02/0.02", "1": "12/0.02/0.02", "2": "22/0.02/0.02", "9": "00/0/0", "12": "00/0/0", "0/0": "00/0/0",
              "1/0": "11/0/0.5", "007": "00/0/0"}
    rows, parts = [], []
    for i in range(N):
        toks = [rng.choice(kinds) for _ in range(M)]
        if i == N - 1:
            toks = toks[:M - 5]                                   # the file ends inside the last line
        rows.append(toks)
        parts.append("i%d" % i)
        for t in toks:
            parts.append(rng.choice(seps))
            parts.append(t)
        parts.append("\r\n" if i < N - 1 else "")
    blob = "".join(parts).encode()
    assert len(blob) > (4 << 20) + (2 << 20)                 # at least one refill inside the tokens
    (tmp_path / "t.gen").write_bytes(blob)
    _, _, inds = parse(exe, tmp_path)
    n51 = 0
    for i, toks in enumerate(rows):
        got = inds["i%d" % i]["geno"]
        assert len(got) == M
        for x, t in enumerate(toks):
            if t == "5/1":
                a, s1, s2 = got[x].split("/")
                assert a == "11" and abs(float(s1) - 0.00212119) < 1e-7 and abs(float(s2) - 0.47464) < 1e-5, (i, x, got[x])
                n51 += 1
            else:
                assert got[x] == expect[t], (i, x, t, got[x])
    assert n51 > 100000
    # past the end of the file: sscanf of an empty token reads nothing, `data` stays at the last token's value and data2 is 0
    last = rows[-1][-1]
    carried = int(last.split("/")[0])
    tail = inds["i%d" % (N - 1)]["geno"][M - 5:]
    assert len(set(tail)) == 1
    assert (tail[0] == "00/0/0") == (carried == 0)
