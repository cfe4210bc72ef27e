"""The parameter-side kernels of the FFT filter (coefficient table and its gradient: lstep_fft_coef_fwd / _bwd) in isolation, T = 100, P = 172:
they sit on the critical chain of every configuration (start and end of every training iteration).
usage: python tools/fftcoef_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat

dev = torch.device("cuda", 0)
nat.LIB_PATH = os.environ.get("LSTEP_LIB", nat.LIB_PATH)
if os.environ.get("LSTEP_ABI"):          # (A/B against a library of an earlier commit)
    nat.ABI_VERSION = int(os.environ["LSTEP_ABI"])
lib = nat.load_library()
T, P = 100, 172
g = torch.Generator(device=dev); g.manual_seed(0)
wr = torch.randn(T, P, 2, device=dev, generator=g)
a = torch.randn(T, device=dev, generator=g)
m = torch.ones(T, dtype=torch.float64, device=dev)
coef = torch.empty(T, P, device=dev); c = torch.empty(T, 2, dtype=torch.float64, device=dev)
gc = torch.randn(T, P, device=dev, generator=g)
g_w = torch.empty(T, P, 2, device=dev); g_a = torch.empty(T, device=dev); scratch = torch.empty(T, 2, dtype=torch.float64, device=dev)


def fwd():
    nat.check(lib.lstep_fft_coef_fwd(nat.ptr(wr), nat.ptr(a), nat.ptr(m), T, P, nat.ptr(coef), nat.ptr(c), nat.current_stream()))


def bwd():
    nat.check(lib.lstep_fft_coef_bwd(nat.ptr(gc), nat.ptr(wr), nat.ptr(c), nat.ptr(m), T, P, nat.ptr(g_w), nat.ptr(g_a), nat.ptr(scratch), nat.current_stream()))


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fwd(); bwd(); torch.cuda.synchronize()
print(f"fft_coef fwd {timed(fwd):6.1f} us   bwd (two kernels) {timed(bwd):6.1f} us   checksum {coef.double().sum().item():.9e} {g_w.double().sum().item():.9e} {g_a.double().sum().item():.9e}")
