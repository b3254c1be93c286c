"""how long tiny kernels + a D2H copy on a side stream take while a render saturates the GPU"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fireflies_amd import mi, workloads
wl = workloads.vocalfold(device="cuda", entity_device="cpu")
with torch.no_grad():
    wl.params["tex.data"] = workloads.build_texture(wl).contiguous()
lo = torch.zeros(3, device="cuda"); hi = torch.ones(3, device="cuda")
pin = torch.empty(64, dtype=torch.float32).pin_memory()
for prio in (0, -1):
    for nstreams in (1, 4):
        sides = [torch.cuda.Stream(priority=prio) for _ in range(nstreams)]
        for busy in (False, True):
            t_k = t_c = t_p = 0.0
            n = 30
            for i in range(n):
                if busy:
                    mi.render(wl.mi_scene, spp=64, seed=i)
                t0 = time.perf_counter()
                outs = []
                for j in range(10):
                    with torch.cuda.stream(sides[j % nstreams]):
                        outs.append(torch.rand(3, device="cuda"))
                for s in sides[1:]:
                    sides[0].wait_stream(s)
                with torch.cuda.stream(sides[0]):
                    flat = torch.cat(outs)
                    ev = torch.cuda.Event(); ev.record()
                    t1 = time.perf_counter()
                    ev.synchronize()
                    t2 = time.perf_counter()
                    v = flat.cpu()
                    t3 = time.perf_counter()
                    pin[:30].copy_(flat, non_blocking=True); ev.record(); ev.synchronize()
                    t4 = time.perf_counter()
                t_k += t2 - t0; t_c += t3 - t2; t_p += t4 - t3
                torch.cuda.synchronize()
            print(f"prio={prio} streams={nstreams} busy={busy}: launch+kernels {t_k/n*1e6:.0f} us, .cpu() {t_c/n*1e6:.0f} us, pinned copy {t_p/n*1e6:.0f} us")
