"""Self-attention backward alone (tld_debug_attention_bwd) at the training shape: HIP-event time per launch and, with --check, the error against
torch autograd on the same bf16-rounded operands.   tools/attn_bwd_bench.py [--ntok 256] [--batch 128] [--heads 12] [--iters 20] [--check]
TLD_LIB=<other .so> selects another build (attribution builds: make EXTRA=-DTLD_AB_DBG=<bits>, see tld_train_attn.hip)."""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_latent_diffusion_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--ntok", type=int, default=256)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
L = _lib.lib()
dev = torch.device("cuda:0")
B, N, H = a.batch, a.ntok, a.heads
d = 64 * H
g = torch.Generator().manual_seed(3)
q, k, v = (torch.randn(B, N, d, generator=g).bfloat16() for _ in range(3))
go = torch.randn(B, N, d, generator=g) * 0.1
sp = lambda t: t.view(B, N, H, 64).transpose(1, 2)
qd, kd, vd = (t.to(dev).float().requires_grad_(True) for t in (q, k, v))
o = torch.nn.functional.scaled_dot_product_attention(sp(qd), sp(kd), sp(vd)).transpose(1, 2).reshape(B, N, d)
qk = torch.cat([q, k], dim=-1).to(dev).contiguous()
vt = v.view(B, N, H, 64).permute(0, 2, 3, 1).contiguous().to(dev)
ob = o.detach().bfloat16().contiguous()
gd = go.to(dev).contiguous()
out = torch.zeros(B * N, 3 * d, dtype=torch.bfloat16, device=dev)
scratch = torch.zeros(2 * B * H * N, dtype=torch.float32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
run = lambda: _lib.check(L.tld_debug_attention_bwd(C.c_void_p(qk.data_ptr()), C.c_void_p(vt.data_ptr()), C.c_void_p(ob.data_ptr()), C.c_void_p(gd.data_ptr()),
                                                    C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()), B, N, H, st), "attention_bwd")
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
flops = 7 * 2.0 * N * N * 64 * B * H
line = f"attention backward B={B} H={H} N={N}: {us:.1f} us / launch, {flops / us / 1e6:.1f} TFLOP/s (7 products)"
if a.check:
    o.backward(gd)
    got = out.float().view(B, N, 3, d)
    errs = [float(((got[:, :, i] - r.grad).norm() / r.grad.norm())) for i, r in enumerate((qd, kd, vd))]
    line += " | rel err dq %.2e dk %.2e dv %.2e" % tuple(errs)
print(line)
