"""Phase timestamps (100 MHz wall clock) of the level-0 quadtree of frame 0; needs a library built with -DORBFE_QT_TIMING
(ORBFE_LIB=ab/liborbfe_qt.so python tools/qt_times.py)."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import capi, synth
W, H = 1241, 376
B = 64
dev = torch.device("cuda:0")
stride = (W + 63) // 64 * 64
host = np.zeros((B, H, stride), np.uint8)
for i in range(B):
    host[i, :, :W] = synth.textured_image(W, H, 100 + i % 16)
frames = torch.from_numpy(host).to(dev)
cap = 2100
kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
ex = capi.ORBextractor(2000, 1.2, 8, 20, 7, device=0)
for _ in range(3):
    ex.extract_batch_device(frames.data_ptr(), B, W, H, stride, stride * H, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, sync=True)
L = capi.lib()
buf = (ctypes.c_ulonglong * 256)()
L.orbfe_debug_qt_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.orbfe_debug_qt_times(buf, 256) == 0
t = list(buf)
t0 = t[0]
print("n candidates", t[5], "passes", t[2])
print("setup %.2f us" % ((t[1] - t0) / 100))
for it in range(int(t[2])):
    b = 8 + it * 8
    info = t[b + 7]
    names = ["hist", "rank", "perParent+scan", "cidx", "stay-scan", "relabel", "write-gen"]
    parts = ["%s %.2f" % (names[j], ((t[b + j + 1] if j < 6 else 0) - t[b + j]) / 100) for j in range(6)]
    print("pass %d Lc=%d sized=%d: " % (it, info >> 32, info & 1) + ", ".join(parts))
print("passes end %.2f us, best+out %.2f us, total %.2f us" % ((t[3] - t0) / 100, (t[4] - t[3]) / 100, (t[4] - t0) / 100))
