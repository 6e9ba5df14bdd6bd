"""Concept check: one dense block's forward chain (5 dependent F(4,3) convs, Cin 64..320) repeated R times --
(a) whole batch on one stream, (b) the two halves of the batch on two streams (independent dependency chains that share the CUs).
Run with SRK_WINO4_NH=1 (4-wave workgroups, two per CU) and SRK_WINO4_NH=2."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = importlib.import_module("super-resolution_amd")._lib
N, H, W, F, R = 32, 64, 64, 64, int(os.environ.get("R", 40))
wps = []
for k in range(1, 6):
    w = torch.randn(F, k * F, 3, 3, device="cuda") * 0.02
    wp = torch.empty(L.packed_floats(k * F, F, 5), device="cuda")
    t = L.PackTable(w.device, 5); t.add(w, wp, M=F, k_off=0, k_len=k * F, K_total=k * F); t.run()
    wps.append(wp)
b = torch.zeros(F, device="cuda")

def chain(D, n):
    for _ in range(R):
        for k in range(1, 6):
            L.conv3x3(L.View(D, 0, k * F), wps[k - 1], b, L.View(D, (k % 5) * F if k < 5 else 0, F) if False else L.View(D, k * F if k < 5 else 0, F),
                      N=n, H=H, W=W, OH=H, OW=W, Cin=k * F, Cout=F, slope=0.01, wp_format=5)

def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)

Dfull = torch.randn(N, H, W, 5 * F, device="cuda") * 0.1
Da, Db = Dfull[:N // 2].clone(), Dfull[N // 2:].clone()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def one_stream():
    chain(Dfull, N)

def two_streams():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    # interleave the host-side launches so that both queues stay fed
    for _ in range(R):
        for k in range(1, 6):
            for s, D in ((sa, Da), (sb, Db)):
                with torch.cuda.stream(s):
                    L.conv3x3(L.View(D, 0, k * F), wps[k - 1], b, L.View(D, k * F if k < 5 else 0, F), N=N // 2, H=H, W=W, OH=H, OW=W,
                              Cin=k * F, Cout=F, slope=0.01, wp_format=5)
    cur.wait_stream(sa); cur.wait_stream(sb)

for name, fn in (("one stream, batch 32", one_stream), ("two streams, 2 x batch 16", two_streams)):
    fn(); fn()
    ms = min(timed(fn) for _ in range(3))
    print("NH=%s  %-28s %8.2f ms for %d convs = %6.1f us per full-batch conv" % (os.environ.get("SRK_WINO4_NH", "2"), name, ms, 5 * R, ms * 1e3 / (5 * R)))
