"""Development: in-kernel phase stamps of the v2 backward kernels (instance 0, last step processed).
Build the instrumented library first:  make -C ddp_pinocchio_amd/csrc stamps   (-> build_ab/libddp_hip_stamps.so)"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddp_pinocchio_amd import capi
capi.LIB_PATH = os.path.join(ROOT, "build_ab", "libddp_hip_stamps.so")
sys.argv = ["x", "--batch", "64", "--reps", "1"]
os.environ.setdefault("DDP_HIP_BWD_GROUPS", "1")
exec(open(os.path.join(ROOT, "tools", "dev_bwd_timing.py")).read())
out = (C.c_ulonglong * 32)()
assert capi.lib().ddp_hip_debug_stamps(out) == 0
v = list(out)
for title, idx, names in (("K5 bwd_dense2", [0, 1, 2, 4], ["load V, F", "Q_x | Q_u, W = V F", "D = F^T W + epilogue"]),
                          ("K4' bwd_gains2", range(8, 13), ["load Q", "Cholesky", "substitutions + gain stores", "V_x, V_xx"])):
    idx = list(idx)
    print(title)
    for k, nm in enumerate(names):
        print(f"  {nm:32s} {(v[idx[k + 1]] - v[idx[k]]) / 100.0:8.2f} us")
    print(f"  {'total':32s} {(v[idx[-1]] - v[idx[0]]) / 100.0:8.2f} us")
print("K4' substitutions (lane 64): rhs load %.2f  forward %.2f  backward %.2f  gain stores %.2f us" % (
    (v[16] - v[10]) / 100.0, (v[17] - v[16]) / 100.0, (v[18] - v[17]) / 100.0, (v[19] - v[18]) / 100.0))
