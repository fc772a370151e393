"""Host-side rules either side of the path, pinned to reference-generated fixtures (tools/gen_golden.py):
FASTA header -> id (tests/golden/ids.json, scripts/id_parser.py:18-100) through the native scanner and the oracle,
and the transform_kmers index tables (tests/golden/transform.npz, scripts/transform_kmers.py:21-88).  CPU only:
the shared library is loaded, no device is touched."""
import random

import numpy as np
import pytest

from tests import helpers


def _answer(fn, header):
    try:
        return fn(header), None
    except Exception as e:   # noqa: BLE001 -- the exception type is part of the contract
        return None, type(e).__name__


def test_id_rules_match_the_reference_fixture():
    from oracle import oracle
    from phamers_amd import id_parser
    cases = helpers.load_json("ids.json")["cases"]
    assert len(cases) >= 30
    for c in cases:
        want = (c["id"], c["error"])
        assert _answer(oracle.get_id, c["header"]) == want, c
        assert _answer(id_parser.get_id, c["header"]) == want, c


def test_native_id_scanner_agrees_with_the_oracle_on_random_headers():
    from oracle import oracle
    from phamers_amd import id_parser
    rnd = random.Random(5)
    pieces = ["_ID_", "ID", "_", "|", ">", " ", "\t", ".", "1", "23", "NC", "x", "-circular", "e5", "inf", "nan", "0x1p3",
              "SuperContig", "length", "gi", "ref", "A.1", "-", "+"]
    for _ in range(4000):
        h = "".join(rnd.choice(pieces) for _ in range(rnd.randint(0, 9)))
        assert _answer(id_parser.get_id, h) == _answer(oracle.get_id, h), repr(h)


def test_fasta_reader_ids(tmp_path):
    """The reader applies the id rules to record.id (first word of the title) on its worker threads; a header
    with none of the shapes raises IndexError like the reference's get_fasta_ids."""
    from phamers_amd import _lib
    good = [c for c in helpers.load_json("ids.json")["cases"] if c["error"] is None and c["header"].strip()
            and not any(ch.isspace() for ch in c["header"].strip())]
    path = tmp_path / "ids.fa"
    with open(path, "w") as f:
        for i, c in enumerate(good):
            f.write(">%s some description %d\nATGCATGC\n" % (c["header"].strip().lstrip(">"), i))
    fa = _lib.Fasta(str(path), threads=3)
    from oracle import oracle
    want = [oracle.get_id(c["header"].strip().lstrip(">")) for c in good]
    assert list(fa.phamers_ids()) == want
    fa.close()
    bad = tmp_path / "bad.fa"
    bad.write_text(">SuperContig_1_ID_1\nATGC\n>plain_contig_name\nATGC\n")
    fb = _lib.Fasta(str(bad))
    with pytest.raises(IndexError):
        fb.phamers_ids()
    fb.close()


def test_transform_tables_match_the_reference_fixture():
    from oracle import oracle
    from phamers_amd import transform_kmers as tk
    z = helpers.load_npz("transform.npz")
    for k in (2, 3, 4):
        for name, (rev, comp) in (("rev", (True, False)), ("comp", (False, True)), ("revcomp", (True, True))):
            want = z["%s_idx_k%d" % (name, k)]
            assert np.array_equal(tk.reference_indices(k, rev, comp), want), (name, k)
            assert np.array_equal(oracle.reference_transform_indices(k, rev, comp), want), (name, k)
            assert np.array_equal(oracle.transform_kmers(z["in_k%d" % k], rev, comp), z["%s_k%d" % (name, k)])
            perm = tk.exact_indices(k, rev, comp)
            assert sorted(perm) == list(range(4 ** k))
    # the reference's tables are not permutations (the documented deviation of exact=True)
    assert len(set(z["revcomp_idx_k4"])) == 64
    with pytest.raises(IndexError):
        oracle.transform_kmers(np.zeros((1, 4 ** 5), dtype=np.int64), True, True)
