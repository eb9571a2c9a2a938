import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seed-story_amd"))
import torch
from seedstory import _lib
l = _lib.lib()
l.ss_debug_tr_probe.argtypes = [C.c_void_p] * 4
src = torch.arange(4096, dtype=torch.int16, device="cuda")
off = []
for lane in range(64):
    g, l15 = lane >> 4, lane & 15
    key = g * 4 + l15 // 4
    off.append(key * 64 + (l15 % 4) * 4)
offt = torch.tensor(off, dtype=torch.int32, device="cuda")
dst = torch.zeros(256, dtype=torch.int16, device="cuda")
l.ss_debug_tr_probe(src.data_ptr(), dst.data_ptr(), offt.data_ptr(), None)
torch.cuda.synchronize()
d = dst.cpu().view(64, 4).tolist()
ok = True
for lane in range(64):
    g, l15 = lane >> 4, lane & 15
    exp = [(g * 4 + j) * 64 + l15 for j in range(4)]
    if d[lane] != exp:
        ok = False
    if lane in (0, 1, 5, 16, 17, 37, 63):
        print(lane, "got", d[lane], "(key,d)=", [(v // 64, v % 64) for v in d[lane]], "expected", exp)
print("hypothesis lane i -> column i of its group's [4 keys][16 d] block:", ok)
