"""stem kernels alone at the BASELINE size (B=16, 512x640, 32 filters, uint8 in, bf16 out): rocprofv3 --kernel-trace --stats -- python tools/stem_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "double-yolo-kaist_amd")]
import torch
from dyk import lib as L
lib = L.load()
B, H, W, cout = 16, 512, 640, 32
img = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device="cuda")
wt = torch.randn(27, cout, device="cuda")
y = torch.zeros(B, H, W, 32, dtype=torch.bfloat16, device="cuda")
dy = torch.randn(B, H, W, 32, device="cuda").bfloat16()
stats = torch.zeros(32, 64, dtype=torch.float64, device="cuda")
d = L.DykStemDesc()
d.img, d.in_u8, d.dtype = img.data_ptr(), 1, L.DYK_BF16
d.B, d.H, d.W, d.Cout, d.k, d.stride, d.pad, d.Ho, d.Wo = B, H, W, cout, 3, 1, 1, H, W
d.wt, d.y, d.ldy, d.stats, d.stats_slots = wt.data_ptr(), y.data_ptr(), 32, stats.data_ptr(), 32
planes = lib.dyk_stem_wgrad_planes(ctypes.byref(d))
part = torch.empty(planes * cout * 27, device="cuda")
dw = torch.zeros(cout * 27, device="cuda")
d.dy, d.lddy, d.dw, d.part = dy.data_ptr(), 32, dw.data_ptr(), part.data_ptr()
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    L.check(lib.dyk_stem_conv_fwd(ctypes.byref(d), s))
    L.check(lib.dyk_stem_conv_wgrad(ctypes.byref(d), s))
torch.cuda.synchronize()
print("planes", planes)
