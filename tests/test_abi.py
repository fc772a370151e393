"""CPU-only: the C-ABI shared library loads and exports every symbol that
include/phamers_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

from tests.helpers import REPO


def declared_functions():
    text = open(os.path.join(REPO, "include", "phamers_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("phk_count_ascii", "phk_normalize_i64", "phk_score", "phk_count_dev", "phk_count_score_dev",
                 "phk_model_create", "phk_synth_packed_dev", "phk_profile_get"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from phamers_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), "missing export: " + name
        assert name in _lib.SIGNATURES, "no ctypes signature for " + name
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.phk_abi_version() == 2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from phamers_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.load()


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, "phamers_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_build_entry_checks_the_version_the_binding_expects():
    """__graft_entry__.build() asserts the library's ABI version: against the binding's constant, not a literal (round 5: the
    bump to 2 left a literal 1 there, found by building a fresh clone)."""
    import inspect

    import __graft_entry__ as entry
    src = inspect.getsource(entry.build)
    assert "_lib.ABI_VERSION" in src and "phk_abi_version() == 1" not in src
    entry.build()
