import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
os.chdir("/root/repo")
import test_hip_parity as T
from axial_vs_amd import _lib
for o in (0, 2, 3, 4):
    _lib.check(_lib.lib().axvs_set_option(b"msda_gemm", o), "opt")
    print("msda_gemm", o)
    try:
        T.test_within_clip_module_full_size_golden()
    except AssertionError as e:
        print("  FAIL", str(e)[:80])
