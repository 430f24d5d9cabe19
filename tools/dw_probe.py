"""Depthwise-conv kernels at the MobileNetV3 cfg's shapes (B=32, 512x640 input): achieved GB/s against in+out bytes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "double-yolo-kaist_amd"))
import torch
from dyk import ops
B = int(os.environ.get("B", "32"))
for (C, H, W, k, s) in [(16, 256, 320, 3, 1), (64, 256, 320, 3, 2), (72, 128, 160, 3, 1), (72, 128, 160, 5, 2), (120, 64, 80, 5, 1),
                        (240, 64, 80, 3, 2), (200, 32, 40, 3, 1), (480, 32, 40, 3, 1), (672, 32, 40, 5, 1), (672, 32, 40, 5, 2), (960, 16, 20, 5, 1)]:
    ld = (C + 31) // 32 * 32
    x = torch.randn(B, H, W, ld, device="cuda").bfloat16()
    w = torch.randn(k * k, C, device="cuda")
    pad = k // 2
    def t(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    y = ops.dwconv_fwd(x, w, k, s, pad, C=C)
    ms_f = t(lambda: ops.dwconv_fwd(x, w, k, s, pad, C=C))
    ms_alloc = t(lambda: torch.zeros_like(y))
    ms_d = t(lambda: ops.dwconv_dgrad(y, w, k, s, pad, H, W, C=C))
    # the weight gradient the way the plan launches it: descriptor built once, partial planes, no allocation in the loop
    import ctypes
    from dyk import lib as L
    lib = L.load()
    dwd = ops._dw_desc(x, y, None, k, s, pad, C)
    dw = torch.zeros(k * k, C, device="cuda")
    rows = lib.dyk_dwconv_wgrad_rows(ctypes.byref(dwd))
    part = torch.zeros(rows, k * k, C, device="cuda")
    dwd.dw, dwd.part = dw.data_ptr(), part.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    ms_w = t(lambda: lib.dyk_dwconv_wgrad(ctypes.byref(dwd), st), n=20)
    by = (x.numel() + y.numel()) * 2
    print("C %4d %3dx%-3d k%d s%d: fwd %.1f us (%.0f GB/s; zeros %.1f us) dgrad %.1f us wgrad %.1f us (%d planes)  [bytes %.0f MB]" % (
        C, H, W, k, s, ms_f * 1e3, by / ms_f / 1e6, ms_alloc * 1e3, ms_d * 1e3, ms_w * 1e3, rows, by / 1e6))
