"""CPU, build container only: the numeric tables the oracle and the kernels carry are compared
entry by entry with the literals in the reference's source text.  Skipped where /root/reference is
absent (the GPU box); reading the reference as text is study, nothing is imported or executed."""
import os
import re

import numpy as np
import pytest

REF = "/root/reference/internal/entropy"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference source not present")


def test_mq_state_table(oracle):
    txt = open(os.path.join(REF, "mqc.go")).read()
    body = re.search(r"var mqStates = \[\]mqState\{(.*?)\n\}", txt, re.S).group(1)
    rows = re.findall(r"\{(0x[0-9A-Fa-f]+),\s*(\d+),\s*(\d+),\s*(\d+)\}", body)
    assert len(rows) == 94
    qe, nm, nl = oracle.mq_table()
    for i, (q, mps, a, b) in enumerate(rows):
        assert (int(q, 16), int(mps), int(a), int(b)) == (int(qe[i]), i & 1, int(nm[i]), int(nl[i])), i


def test_ht_tables_match_generated_headers():
    txt = open(os.path.join(REF, "ht_luts.go")).read()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for hdr in ("oracle/ht_tables.h", "go-jpeg2000_amd/csrc/ht_tables.h"):
        h = open(os.path.join(root, hdr)).read()
        for name, macro in (("vlcTbl0", "J2K_HT_VLC_TBL0"), ("vlcTbl1", "J2K_HT_VLC_TBL1")):
            ref = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", re.search(r"var %s = \[1024\]uint16\{(.*?)\n\}" % name, txt, re.S).group(1))]
            got = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{4}", re.search(r"#define %s_INIT \{(.*?)\n\}" % macro, h, re.S).group(1))]
            assert got == ref + [0] * (1024 - len(ref))


def test_context_constants():
    txt = open(os.path.join(REF, "mqc.go")).read()
    names = re.findall(r"^\s*(Ctx[A-Za-z0-9]+)", re.search(r"const \(\s*// Zero coding.*?NumContexts", txt, re.S).group(0), re.M)
    assert names.index("CtxSC0") == 9 and names.index("CtxMag0") == 14 and names.index("CtxRL") == 17 and names.index("CtxUni") == 18
