#!/usr/bin/env python
"""Stage-by-stage check of the HIP CLIP text tower on a one-block configuration (debugging aid)."""
import ctypes as C, math, os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from transformer_latent_diffusion_amd import _lib
from transformer_latent_diffusion_amd.clip_text import ClipTextConfig, ClipTextEncoder, synth_clip_state_dict
from test_clip_host import _tokens

cfg = ClipTextConfig(vocab_size=1000, context_length=16, width=128, heads=2, layers=1, embed_dim=64)
sd = synth_clip_state_dict(cfg, 3)
enc = ClipTextEncoder(cfg, max_batch=4)
enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
enc.to("cuda")
text = _tokens(cfg, 3, 1)
got = enc.encode_text(text.cuda()).cpu()
w = {k: torch.from_numpy(v) for k, v in sd.items()}
b, n, d, h = 3, 16, 128, 2
x0 = w["token_embedding.weight"][text] + w["positional_embedding"]
p = "transformer.resblocks.0."
y = F.layer_norm(x0, (d,), w[p + "ln_1.weight"], w[p + "ln_1.bias"])
qkv = F.linear(y, w[p + "attn.in_proj_weight"], w[p + "attn.in_proj_bias"])
q, k, v = qkv.chunk(3, -1)
sp = lambda t: t.view(b, n, h, 64).transpose(1, 2)
s = sp(q) @ sp(k).transpose(-1, -2) / 8 + torch.full((n, n), float("-inf")).triu_(1)
att = (torch.softmax(s, -1) @ sp(v)).transpose(1, 2).reshape(b, n, d)
x1 = x0 + F.linear(att, w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])
hh = F.layer_norm(x1, (d,), w[p + "ln_2.weight"], w[p + "ln_2.bias"])
f = F.linear(hh, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"]); f = f * torch.sigmoid(1.702 * f)
tmp = F.linear(f, w[p + "mlp.c_proj.weight"])
x2 = x1 + tmp + w[p + "mlp.c_proj.bias"]
xf = F.layer_norm(x2, (d,), w["ln_final.weight"], w["ln_final.bias"])
pooled = xf[torch.arange(b), text.argmax(-1)]
out = pooled @ w["text_projection"]
L = _lib.lib()
def rd(name, shape):
    a = np.empty(shape, np.float32)
    _lib.check(L.tld_clip_read_buffer(enc._engine, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size), name)
    return torch.from_numpy(a)
rel = lambda a, r: float((a - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
T = b * n
for name, ref, shape in (("qkv", qkv, (T, 3 * d)), ("att", att, (T, d)), ("x", x1, (T, d)), ("h", hh, (T, d)), ("f", f, (T, 4 * d)),
                         ("tmp", tmp, (T, d)), ("pooled", pooled, (b, d))):
    print(f"{name:7s} rel-rms {rel(rd(name, shape), ref.reshape(shape)):.3e}")
print(f"out     rel-rms {rel(got, out):.3e}")
