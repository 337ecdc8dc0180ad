"""r06: what the vendor's fp32 GEMM (rocBLAS / hipBLASLt through torch.bmm) does on the three recompute products of the fused
correlation's backward (dvc_amd/corr_autograd.py), beside this library's 1x1-convolution engine on the same operands."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd"))
import torch  # noqa: E402

from dvc_amd import ops  # noqa: E402

dev = torch.device("cuda")
B, C, P, R = 16, 256, 5184, 2048
h, w = 54, 96
g = torch.Generator().manual_seed(0)
theta_blk = torch.randn(B, C, R, generator=g).to(dev)
phi = torch.randn(B, C, P, generator=g).to(dev)
dS = torch.randn(B, R, P, generator=g).to(dev)


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("torch", torch.__version__, "allow_tf32", torch.backends.cuda.matmul.allow_tf32, "precision", torch.get_float32_matmul_precision())
fl = 2.0 * B * R * P * C
outF = torch.empty(B, R, P, device=dev)
t = timeit(lambda: torch.bmm(theta_blk.transpose(1, 2), phi, out=outF))
print(f"F = theta_blk^T phi      [B,{R},{C}] x [B,{C},{P}]: torch.bmm {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s")
ref = torch.bmm(theta_blk[:1].double().transpose(1, 2), phi[:1].double())
print("   max rel err vs fp64:", ((outF[:1].double() - ref).abs().max() / ref.abs().max()).item())
tb = theta_blk.view(B, C, 1, R).contiguous()
Fb = torch.empty(B, R, h, w, device=dev)
t = timeit(lambda: ops.conv2d(phi.view(B, C, h, w), tb, None, ksize=1, pad=0, out=Fb))
print(f"   1x1 engine {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s; max rel err vs fp64:", ((Fb[:1].view(1, R, P).double() - ref).abs().max() / ref.abs().max()).item())
dphi = torch.zeros(B, C, P, device=dev)
t = timeit(lambda: torch.baddbmm(dphi, theta_blk, dS, out=dphi))
print(f"dphi += theta_blk dS     [B,{C},{R}] x [B,{R},{P}]: torch.baddbmm {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s")
tbt = theta_blk.transpose(1, 2).contiguous().view(B, R, 1, C)
dphi_img = torch.zeros(B, C, h, w, device=dev)
t = timeit(lambda: ops.conv2d(dS.view(B, R, h, w), tbt, None, ksize=1, pad=0, residual=dphi_img, out=dphi_img))
print(f"   1x1 engine {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s")
dth = torch.empty(B, C, R, device=dev)
t = timeit(lambda: torch.bmm(phi, dS.transpose(1, 2), out=dth))
print(f"dtheta = phi dS^T        [B,{C},{P}] x [B,{P},{R}]: torch.bmm (transposed operand) {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s")
dST = dS.transpose(1, 2).contiguous().view(B, P, R // 32, 32)
phi_t = phi.transpose(1, 2).contiguous().view(B, P, 1, C)
t = timeit(lambda: ops.conv2d(dST, phi_t, None, ksize=1, pad=0))
print(f"   1x1 engine (on a transposed copy of dS) {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s")
