"""lstep_linear_wgrad while another stream keeps the memory system busy (a 1.4 GB device copy, like the engine's snapshot clone): the products
must not change.  (They did with the first, inline-asm software pipeline of the kernel: exact alone, garbage under contention.)
usage: python tools/wgrad_race.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat
dev = torch.device("cuda", 0)
torch.manual_seed(0)
big_a = torch.randn(350_000_000, device=dev)
big_b = torch.empty_like(big_a)
side = torch.cuda.Stream()
for (m, n, k) in ((16384, 176, 176), (49152, 176, 272), (49152, 272, 272), (32768, 176, 176)):
    dy = torch.randn(m, n, device=dev) * 1e-6
    x = torch.randn(m, k, device=dev)
    ref = dy.double().t() @ x.double()
    scale = ref.abs().max().item()
    for mode in ("alone", "with copy"):
        bad = 0
        worst = 0.0
        for rep in range(20):
            torch.cuda.synchronize()
            if mode == "with copy":
                with torch.cuda.stream(side):
                    big_b.copy_(big_a, non_blocking=True)
            dw, db = nat.linear_wgrad(dy, x)
            torch.cuda.synchronize()
            err = (dw.double() - ref).abs().max().item()
            worst = max(worst, err)
            bad += err > 1e-3 * scale
        print(f"m={m} n={n} k={k} {mode:10s}: bad {bad}/20 worst err {worst:.3e} (scale {scale:.3e})", flush=True)
