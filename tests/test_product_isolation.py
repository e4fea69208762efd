"""Host-logic tests that need no GPU: the C-ABI library loads and exports every declared symbol,
the product package never touches the oracle, and a missing library fails loudly."""
import os
import re

import pytest

from gfdl_atmos_cubed_sphere_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_match_export_list():
    hdr = open(os.path.join(ROOT, "include", "fv3_mi355x.h")).read()
    declared = set(re.findall(r"\b(fv3_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.EXPORTS)


def test_product_library_exports_every_symbol():
    if not os.path.exists(L.PRODUCT_SO):
        import __graft_entry__ as ge
        ge.build()
    lib = L.Fv3Lib(L.PRODUCT_SO)   # checks every symbol in EXPORTS (no compute call)
    assert lib.dll.fv3_last_error is not None


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(L.Fv3Error):
        L.Fv3Lib(str(tmp_path / "nope.so"))


def test_package_never_imports_oracle_or_hostemu():
    pkg = os.path.join(ROOT, "gfdl_atmos_cubed_sphere_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle_lib" not in src and "libfvo" not in src and "fvo.h" not in src, f
                assert "libfv3_hostemu" not in src, f
