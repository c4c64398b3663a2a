"""Clocks and package power (rocm-smi) while the fp32 expand + reduce chain kernel runs in a loop: is the kernel at the chip's power
limit?  python tools/probes/chain_clock_probe.py   (MI355X: idle 299 W / busy 1376 - 1382 W of 1400, sclk 2260 MHz)"""
import ctypes, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindpose_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
n, h, w = 128, 64, 48
mid, res = torch.randn(n, 64, h, w, device=dev), torch.randn(n, 256, h, w, device=dev)
w3, w1 = torch.randn(256, 64, 1, 1, device=dev) * 0.17, torch.randn(64, 256, 1, 1, device=dev) * 0.09
s3, b3, s1, b1 = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev), torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
y, z = torch.empty(n, 256, h, w, device=dev), torch.empty(n, 64, h, w, device=dev)
pk3 = torch.empty(lib.mp_conv_packed_weight_bytes(256, 64, 1, 1) // 4, device=dev)
pk1 = torch.empty(lib.mp_conv_packed_weight_bytes(64, 256, 1, 1) // 4, device=dev)
_lib.check(lib.mp_conv_pack_weight(_lib.ptr(w3), _lib.ptr(pk3), 256, 64, 1, 1, 0, 0, 0, _lib.stream()), "pack")
_lib.check(lib.mp_conv_pack_weight(_lib.ptr(w1), _lib.ptr(pk1), 64, 256, 1, 1, 0, 0, 0, _lib.stream()), "pack")
def chain():
    lib.mp_expand_reduce_fwd(_lib.ptr(mid), _lib.ptr(res), None, None, None, None, _lib.ptr(pk3), _lib.ptr(s3), _lib.ptr(b3), _lib.ptr(pk1), _lib.ptr(s1), _lib.ptr(b1), _lib.ptr(y), _lib.ptr(z), n, 64, 256, 64, h, w, _lib.stream())
samples = []
stop = False
def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip()[:600])
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.5)
t = threading.Thread(target=sample); t.start()
time.sleep(1.5)
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(200): chain()
    torch.cuda.synchronize()
stop = True; t.join()
print("idle:", samples[0]); print("busy:", samples[len(samples) // 2 + 1] if len(samples) > 3 else samples[-1]); print("busy late:", samples[-2])
