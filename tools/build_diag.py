"""Build the diagnostic variants of libaxvs.so that tools/timeline.sh / tools/stamps.py load through AXVS_LIB_PATH (they are not
tracked and not part of the product):
    tools/diag_stamps.so      -DAXVS_STAMPS                     s_memtime phase stamps in the trajectory kernels
    tools/diag_stamps_ffn.so  -DAXVS_STAMPS -DAXVS_STAMPS_FFN   ... inside the FFN half instead
    tools/diag_stamps_qkv.so  -DAXVS_STAMPS -DAXVS_STAMPS_QKV   ... in the QKV kernel instead
    tools/diag_stamps_tr.so   -DAXVS_STAMPS -DAXVS_STAMPS_TR    ... in the training tier's X W^T GEMM (tools/gemm_stamps.py)
    tools/diag_ablw.so        -DAXVS_ABL_W                      every weight-fragment load hits one L1-resident KiB (wrong results on
                                                                purpose: the time difference is the exposed L2 -> CU weight stream)
Run in the build container (hipcc cross-compiles): python tools/build_diag.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
for name, flags in (("diag_stamps", ["-DAXVS_STAMPS"]), ("diag_stamps_ffn", ["-DAXVS_STAMPS", "-DAXVS_STAMPS_FFN"]),
                    ("diag_stamps_qkv", ["-DAXVS_STAMPS", "-DAXVS_STAMPS_QKV"]), ("diag_ablw", ["-DAXVS_ABL_W"]),
                    ("diag_stamps_tr", ["-DAXVS_STAMPS", "-DAXVS_STAMPS_TR"])):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    ge.build(extra_flags=flags, lib_path=os.path.join(ROOT, "tools", name + ".so"), load=False)
