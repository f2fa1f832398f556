"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol the public headers
declare (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT

INC = os.path.join(ROOT, "include")


def _declared_c_symbols():
    txt = open(os.path.join(INC, "stm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(stm_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_reference_stage_api():
    names = _declared_c_symbols()
    # one host + one device flavour per reference stage wrapper
    for stage in ["ci_adcensus", "ca_cross", "dc_wta", "dc_hslo", "dr_dcc", "dr_irv", "filter_bilateral_1",
                  "filter_gaussian_1", "filter_bleed_1", "filter_median", "dibr_occl", "dibr_occl_to_mask", "dibr_dbm", "dibr_dfm",
                  "mux_multiview", "adcensus_stm"]:
        assert "stm_" + stage in names, stage
        assert "stm_d_" + stage in names, "d_" + stage


def test_library_exports_every_declared_symbol(stm):
    lib = ctypes.CDLL(stm.LIB_PATH)
    missing = [n for n in _declared_c_symbols() if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_prototypes_cover_the_header(stm):
    from stm_amd import _lib
    assert sorted(_lib.PROTOS) == _declared_c_symbols()


def test_dropin_cxx_names_are_exported(stm):
    """stm_dropin.hpp: the reference's own C++ names (mangled) must be defined by the library."""
    txt = open(os.path.join(INC, "stm_dropin.hpp")).read()
    txt = re.sub(r"//.*", "", txt)
    want = sorted(set(re.findall(r"^void\s+([A-Za-z_0-9]+)\s*\(", txt, flags=re.M)))
    assert len(want) == 33
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", stm.LIB_PATH]).decode()
    have = set(re.findall(r" T ([A-Za-z_0-9]+)\(", out))
    assert not [w for w in want if w not in have]


def test_version_call_needs_no_gpu(stm):
    assert stm.lib().stm_version() >= 100


def test_headers_compile_as_plain_c_and_cxx(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "stm_hip.h"\nint main(void){return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", INC, "-c", str(c), "-o", str(tmp_path / "t.o")])
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "stm_dropin.hpp"\n#include "stm_hip.h"\nint main(){return 0;}\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", INC, "-c", str(cpp), "-o", str(tmp_path / "u.o")])
