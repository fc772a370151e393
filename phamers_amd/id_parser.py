"""
id_parser.py -- FASTA header -> id, the rules of PhaMers' scripts/id_parser.py:18-100, evaluated by the native
scanner in csrc/fasta.cpp (phk_parse_id; the FASTA reader applies the same scanner to every record on its worker
threads, see _lib.Fasta.phamers_ids).  This module is the string-in / string-out binding with the reference's
names and failure behaviour: IndexError for a header that has none of the three shapes, None where the
reference's bacteria rule finds no accession.  tests/golden/ids.json holds the reference's answers.
"""
import ctypes

from . import _lib

_OK, _INDEX_ERROR, _NONE = 0, 1, 2


def get_id(header):
    """scripts/id_parser.py:89-100 (contig '_ID_' / phage 'a|b|c|ACC|' / bacteria accession)."""
    raw = header.encode("latin-1", "replace")
    lib = _lib.load()
    out = ctypes.create_string_buffer(len(raw) + 1)
    n, status = ctypes.c_uint64(), ctypes.c_int()
    _lib.check(lib.phk_parse_id(raw, len(raw), out, len(raw) + 1, ctypes.byref(n), ctypes.byref(status)))
    if status.value == _INDEX_ERROR:
        raise IndexError("list index out of range")
    if status.value == _NONE:
        return None
    return out.raw[: n.value].decode("latin-1")
