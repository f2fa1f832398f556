"""Build helper: compiles the HIP C-ABI library (csrc/ -> libstm_hip.so) for gfx950 with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libstm_hip.so")


def build(force=False, jobs=8):
    csrc = os.path.join(HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", csrc, "all", "timing"])
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce %s" % LIB_PATH)
    return LIB_PATH
