"""clock64 ticks per phase of kd_block_kernel (variant -DPCLHIP_KDB_TICKS): PCLHIP_LIB=pcl_amd/variants/libpclhip_kdbticks.so"""
import ctypes, os, sys, torch
sys.path.insert(0, ".")
import pcl_amd.api as A
from pcl_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = A.Context()
tgt = synth.gaussian_surface_device(n, seed=1)
t = A.KdTree(ctx)
t.setInputCloud(tgt)
lib = ctypes.CDLL(os.environ["PCLHIP_LIB"])
out = (ctypes.c_ulonglong * 8)()
lib.pclhip_debug_kdb_ticks(out, 1)
t.setInputCloud(tgt)
torch.cuda.synchronize()
lib.pclhip_debug_kdb_ticks(out, 0)
nb = (n + 4095) // 4096
names = ["load", "boxes+axis (6 levels)", "4way keys+setup (2)", "4way selection passes", "4way classify+scan+permute", "binary levels (4)", "output+rank", "(selection passes per block)"]
tot = sum(out[i] for i in range(7))
for i in range(8):
    print("%-32s %10.0f ticks per block  %5.1f %%" % (names[i], out[i] / nb, 100.0 * out[i] / tot if i < 7 else 0))
print("build %.3f ms" % t.build_ms())
