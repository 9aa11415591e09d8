"""Shader clock (MHz) sampled every ms by a one-wave probe kernel on a side stream while MMVit4 steps run (GPU box)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); import helpers  # noqa
import torch, mmvit4, ops
dev = "cuda:0"
_so = os.path.join(ROOT, "tools", "probe", "clock_probe.so")
if not os.path.exists(_so):       # hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/probe/clock_probe.hip -o tools/probe/clock_probe.so
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", _so[:-3] + ".hip", "-o", _so])
lib = ctypes.CDLL(_so)
lib.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
torch.manual_seed(0)
model = mmvit4.MMVit4().to(dev).train()
x, mask = helpers.make_inputs(32, 4, 224, 224); x, mask = x.to(dev), mask.to(dev)
def step():
    model.zero_grad(set_to_none=True)
    loss = ops.bce_with_logits_mean(model(x), mask); loss.backward()
if os.environ.get("MODE") == "gemm":       # back-to-back 4096^3 GEMMs instead of the model step
    import corrif_hip as H
    A = torch.randn(4096, 4096, device=dev); Bm = torch.randn(4096, 4096, device=dev); Cm = torch.empty(4096, 4096, device=dev)
    def step():
        for _ in range(150): ops.gemm(A.data_ptr(), 4096, Bm.data_ptr(), 4096, 1, Cm.data_ptr(), 4096, 4096, 4096, 4096, 4096, H.gemm_geom())
for _ in range(2): step()
torch.cuda.synchronize()
N = 800
buf = torch.zeros(2 * N, dtype=torch.int64, device=dev)
side = torch.cuda.Stream(device=dev)
assert lib.clock_probe(buf.data_ptr(), N, 100000, side.cuda_stream) == 0
for _ in range(2): step()
torch.cuda.synchronize()
b = buf.cpu().view(N, 2)
w, c = b[:, 0].double(), b[:, 1].double()
mhz = (c[1:] - c[:-1]) / (w[1:] - w[:-1]) * 100.0
t = (w[1:] - w[0]) / 1e5
for i in range(0, N - 1, 10):
    print("t %6.1f ms  %7.1f MHz" % (t[i].item(), mhz[i:i + 10].mean().item()))
