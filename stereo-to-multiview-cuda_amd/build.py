"""Build helper: compiles the HIP C-ABI library (csrc/ -> libstm_hip.so) for gfx950 with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libstm_hip.so")


def build(force=False, jobs=8):
    csrc = os.path.join(HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", csrc, "all"])  # the product: its failure is the build's failure
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce %s" % LIB_PATH)
    # the timing-experiment library (tools/*_time.py only; its kernels can skip work and give invalid results) is best effort:
    # a failure there must not take the product build with it
    rc = subprocess.call(["make", "-s", "-j%d" % jobs, "-C", csrc, "timing"])
    if rc != 0:
        import sys
        print("stm_amd.build: libstm_hip_timing.so did not build (make timing -> %d); the product library is unaffected" % rc, file=sys.stderr)
    return LIB_PATH
