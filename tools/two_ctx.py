"""Experiment: one render call vs K concurrent calls (own context each) over disjoint sample ranges of the same frame."""
import importlib, sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("raytracer-rs_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "utah"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w, h = 1920, 1080
hs = rt.HostScene(name, w / h)
for K in (1, 2, 3, 4):
    ctxs = [rt.Context(0) for _ in range(K)]
    dss = [rt.DeviceScene(c, hs.desc) for c in ctxs]
    accs = [torch.zeros(h, w, 3, device="cuda") for _ in range(K)]
    def work(k):
        p = rt.default_params(w, h, spp)
        p.spp_begin = k * spp // K
        p.spp_end = (k + 1) * spp // K
        dss[k].render_device(hs.camera, p, accs[k].data_ptr())
    best = 1e9
    for rep in range(3):
        for a in accs: a.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    tot = sum(accs)
    print(f"{name} {spp} spp, {K} concurrent contexts: {best*1e3:.2f} ms  mean {float(tot.mean())/spp:.6f}")
    del dss, ctxs
