"""Throw-away probe: s_memtime stamps inside one tile of the layer-2 weight-gradient kernel (built with -DLR_TR2_STAMPS)."""
import ctypes, sys, torch
L = ctypes.CDLL("lipreading_amd/_lib/liblr_stamps.so")
P = ctypes.c_void_p
L.lr_conv3d_wgrad_workspace_bytes.restype = ctypes.c_size_t
L.lr_conv3d_wgrad_workspace_bytes.argtypes = [ctypes.c_int] * 5
L.lr_conv3d_wgrad.argtypes = [P, P, P, P, P, ctypes.c_size_t] + [ctypes.c_int] * 15 + [P]
dev = torch.device("cuda:0")
B, T, hw, cin, cout, k = 32, 75, 24, 32, 64, (3, 5, 5)
x = torch.randn(B * T, hw, hw, cin, device=dev).clamp_min(0).bfloat16()
dz = (torch.randn(B * T, hw, hw, cout, device=dev) * 0.1).bfloat16()
wb = L.lr_conv3d_wgrad_workspace_bytes(cout, cin, *k)
ws = torch.empty(wb, dtype=torch.uint8, device=dev)
dw = torch.empty((cout, cin) + k, device=dev)
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  rc = L.lr_conv3d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), wb, 0, B, T, hw, hw, cin, cin, cout, *k, 1, 1, 2, 2, st)
  e1.record(); torch.cuda.synchronize()
  out = (ctypes.c_ulonglong * 16)()
  L.lr_debug_tr2_stamps(out)
  s = list(out)[:8]
  print("rc", rc, "ms %.3f" % e0.elapsed_time(e1), "segments", [s[i + 1] - s[i] for i in range(7)], "tile", s[7] - s[0])
