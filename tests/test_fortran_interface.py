"""The ISO_C_BINDING interface module compiles with the image's Fortran compiler and
declares a binding for every compute entry point of the C ABI."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOD = os.path.join(ROOT, "gfdl_atmos_cubed_sphere_amd", "fortran", "fv3_mi355x_mod.F90")


def test_bindings_cover_the_header():
    src = open(MOD).read()
    bound = set(re.findall(r'bind\(C, name="(fv3_[a-z0-9_]+)"\)', src))
    hdr = open(os.path.join(ROOT, "include", "fv3_mi355x.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*(fv3_[a-z0-9_]+)\(", hdr, flags=re.M))
    # everything the header declares is bound, except the two profiling helpers and memset (host-tool only)
    missing = declared - bound - {"fv3_profile", "fv3_profile_report", "fv3_memset"}
    assert not missing, missing


def test_module_compiles(tmp_path):
    fc = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"
    if not os.path.exists(fc):
        pytest.skip("no Fortran compiler in this image")
    subprocess.check_call([fc, "-c", MOD, "-o", str(tmp_path / "m.o"), "-module-dir", str(tmp_path)])
