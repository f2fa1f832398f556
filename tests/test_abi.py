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


# The 33 host symbols of the reference's per-stage API, Itanium-mangled from the declarations in its d_*.h headers
# (derived once in the build container with g++; d_io.h:32-52 declares `int angle` for adcensus_stm / adcensus_stm_2).
REFERENCE_HOST_SYMBOLS = """
_Z10d_ca_crossPhPPfS1_S1_S0_PS_ffiiiiii _Z10d_dibr_dbmPhS_S_PfS0_S_S_S0_S0_fiii _Z10d_dibr_dfmPhS_S_PfS0_fiii
_Z10d_tx_scalePhS_iiiii _Z11ci_adcensusPhS_PPfS1_ffiiiii _Z11d_dibr_occlPhS_PfS0_ii
_Z12adcensus_stmPhPfS0_S_iiiiiiiiiiffffiiif _Z13d_ci_adcensusPhS_PPfS1_S1_S1_S0_ffiiiii _Z13filter_medianPfii
_Z13mux_multiviewPPhS_ifiiiii _Z14adcensus_stm_2PhPfS0_S_iiiiiiiifiiiiffffiiif _Z14filter_bleed_1Phiii
_Z15d_filter_medianPfii _Z15d_mux_multiviewPPhS_ifiiiii _Z16d_filter_bleed_1Phiii _Z17dibr_occl_to_maskPfS_PhS0_ii
_Z17filter_gaussian_1Pfifii _Z18filter_bilateral_1Pfiffiii _Z19d_dibr_occl_to_maskPfS_PhS0_ii
_Z19d_filter_gaussian_1Pfifii _Z20d_filter_bilateral_1Pfiffiii _Z22generateGaussianKernelPfif _Z6dc_wtaPPfS_iiii
_Z6dr_dccPhS_PfS0_ii _Z6dr_irvPfPhPS0_ifiiiiii _Z7dc_hsloPPfS_PhS1_fffiiiii _Z8ca_crossPhPS_PPfS2_ffiiiiii
_Z8d_dc_wtaPPfS_iiii _Z8d_dr_dccPhS_PfS0_ii _Z8d_dr_irvPfPhPS0_ifiiiiii _Z8dibr_dbmPhS_S_PfS0_S_S_S0_S0_fiii
_Z8dibr_dfmPhS_S_PfS0_fiii _Z9dibr_occlPhS_PfS0_ii
""".split()


def test_dropin_exports_the_33_reference_symbols(stm):
    """Link-level parity: an object compiled against the reference's own headers resolves against libstm_hip.so."""
    assert len(REFERENCE_HOST_SYMBOLS) == 33 and len(set(REFERENCE_HOST_SYMBOLS)) == 33
    out = subprocess.check_output(["nm", "-D", "--defined-only", stm.LIB_PATH]).decode()
    have = set(re.findall(r" T (_Z\S+)", out))
    assert not [w for w in REFERENCE_HOST_SYMBOLS if w not in have]
    # the fractional-angle additions exist next to them under their own names
    assert any(h.startswith("_Z14adcensus_stm_f") for h in have) and any(h.startswith("_Z16adcensus_stm_2_f") for h in have)


def test_dropin_header_declares_what_the_library_defines(stm, tmp_path):
    """stm_dropin.hpp: every function it declares, compiled by g++, mangles to a symbol the library defines."""
    txt = open(os.path.join(INC, "stm_dropin.hpp")).read()
    txt = re.sub(r"//.*", "", txt)
    want = sorted(set(re.findall(r"^void\s+([A-Za-z_0-9]+)\s*\(", txt, flags=re.M)))
    assert len(want) == 35  # 33 reference names + adcensus_stm_f + adcensus_stm_2_f
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", stm.LIB_PATH]).decode()
    have = set(re.findall(r" T ([A-Za-z_0-9]+)\(", out))
    assert not [w for w in want if w not in have]
    # the int-angle declaration is the one a reference-style call binds to
    cpp = tmp_path / "call.cpp"
    cpp.write_text('#include "stm_dropin.hpp"\nvoid f(unsigned char* a, float* b){ float angle = 18.43f; '
                   'adcensus_stm(a, b, b, a, 1, 2, 1, 1, 1, 3, 8, angle, 16, 8, 10.f, 30.f, 6.f, 20.f, 17, 8, 20, 0.4f); }\n')
    obj = tmp_path / "call.o"
    subprocess.check_call(["g++", "-std=c++11", "-Wno-float-conversion", "-I", INC, "-c", str(cpp), "-o", str(obj)])
    und = subprocess.check_output(["nm", "-u", str(obj)]).decode()
    assert "_Z12adcensus_stmPhPfS0_S_iiiiiiiiiiffffiiif" in und


def test_version_call_needs_no_gpu(stm):
    assert stm.lib().stm_version() >= 100


def test_headers_compile_as_plain_c_and_cxx(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "stm_hip.h"\nint main(void){return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", INC, "-c", str(c), "-o", str(tmp_path / "t.o")])
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "stm_dropin.hpp"\n#include "stm_hip.h"\nint main(){return 0;}\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", INC, "-c", str(cpp), "-o", str(tmp_path / "u.o")])
