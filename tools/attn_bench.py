"""Self-attention forward alone (tld_debug_attention_fwd): error against an fp32 torch evaluation of the same bf16 inputs and the HIP-event
time per launch.  tools/attn_bench.py [--ntok 1024] [--batch 32] [--heads 12] [--iters 20]."""
import argparse, ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_latent_diffusion_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ntok", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--scale", type=float, default=1.0, help="std of q and k (larger = peakier softmax)")
    a = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    B, N, H = a.batch, a.ntok, a.heads
    d = 64 * H
    q = (torch.randn(B, N, H, 64, generator=g) * a.scale).to(torch.bfloat16)
    k = (torch.randn(B, N, H, 64, generator=g) * a.scale).to(torch.bfloat16)
    v = torch.randn(B, N, H, 64, generator=g).to(torch.bfloat16)
    qk = torch.cat([q.reshape(B * N, d), k.reshape(B * N, d)], dim=1).contiguous().to(dev)
    vt = v.permute(0, 2, 3, 1).reshape(B, d, N).contiguous().to(dev)        # [B, H*64, N]
    att = torch.zeros(B * N, d, dtype=torch.bfloat16, device=dev)
    ms = ctypes.c_float(0)
    st = torch.cuda.current_stream().cuda_stream
    rc = L.tld_debug_attention_fwd(qk.data_ptr(), vt.data_ptr(), att.data_ptr(), B, N, H, 2, ctypes.byref(ms), st)
    _lib.check(rc, "tld_debug_attention_fwd")
    rc = L.tld_debug_attention_fwd(qk.data_ptr(), vt.data_ptr(), att.data_ptr(), B, N, H, a.iters, ctypes.byref(ms), st)
    _lib.check(rc, "tld_debug_attention_fwd")
    torch.cuda.synchronize()
    # run-to-run determinism: the same launch into a second buffer, several times
    att2 = torch.empty_like(att)
    same = True
    for _ in range(5):
        att2.zero_()
        _lib.check(L.tld_debug_attention_fwd(qk.data_ptr(), vt.data_ptr(), att2.data_ptr(), B, N, H, 1, None, st), "tld_debug_attention_fwd")
        torch.cuda.synchronize()
        same = same and bool(torch.equal(att, att2))
        if not same:
            bad = (att != att2).nonzero()
            print("MISMATCH rows/cols sample:", bad[:8].tolist(), "count", int(bad.shape[0]))
            break
    # reference on a few samples (fp32 math on the same bf16 values)
    nb = min(B, 2)
    qf = q[:nb].float().to(dev).permute(0, 2, 1, 3); kf = k[:nb].float().to(dev).permute(0, 2, 1, 3); vf = v[:nb].float().to(dev).permute(0, 2, 1, 3)
    ref = torch.softmax(qf @ kf.transpose(-1, -2) / 8.0, dim=-1) @ vf                     # [nb, H, N, 64]
    ref = ref.permute(0, 2, 1, 3).reshape(nb * N, d)
    got = att[: nb * N].float()
    err = (got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()
    if os.environ.get("A2_CLK"):
        ticks = att.view(torch.int64).flatten()[:2].tolist()
        print("clock probe: %d shader ticks / %d ref ticks -> %.0f MHz, workgroup lifetime %.1f us" % (ticks[0], ticks[1], ticks[0] / max(ticks[1], 1) * 100.0, ticks[1] / 100.0))
    flops = 4.0 * B * H * N * N * 64
    print(json.dumps({"ntok": N, "batch": B, "heads": H, "ms": round(ms.value, 4), "pflops": round(flops / (ms.value * 1e-3) / 1e15, 3),
                      "rel_rms_vs_fp32": float(err), "max_abs": float((got - ref).abs().max()), "attn2": os.environ.get("TLD_ATTN2", "1"),
                      "finite": bool(torch.isfinite(att.float()).all()), "deterministic": same}))


if __name__ == "__main__":
    main()
