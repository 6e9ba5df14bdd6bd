import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = importlib.import_module("super-resolution_amd")._lib
lib = L.lib()
N, H, W, F, ci = 32, 64, 64, 64, 128
buf = torch.randn(N, H, W, 320, device="cuda"); out = torch.empty(N, H, W, F, device="cuda")
w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
wp = torch.empty(L.packed_floats(ci, F, 5), device="cuda")
t = L.PackTable(buf.device, 5); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
b = torch.zeros(F, device="cuda")
stamps = torch.zeros(2 * 4096 * 16, dtype=torch.int64, device="cuda")
lib.srk_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
for _ in range(50):
    L.conv3x3(L.View(buf, 0, ci), wp, b, L.View(out), N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F, slope=0.01, wp_format=5)
torch.cuda.synchronize()
nwg = 512 if os.environ.get("SRK_WINO4_NH") == "1" else 256
s = stamps.cpu().view(-1, 16)[:nwg].double() * 0.01
t0 = s[:, 0].min()
st, en = s[:, 0] - t0, s[:, 4] - t0
print("workgroups", nwg, "start: min %.2f median %.2f max %.2f us | end: min %.2f median %.2f max %.2f us" % (st.min(), st.median(), st.max(), en.min(), en.median(), en.max()))
print("started within the first 2 us:", int((st < 2.0).sum()), "of", nwg)
