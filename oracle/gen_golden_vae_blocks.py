"""Writes tests/golden/g18_vae_janus.npz: the VAE decoder's blocks -- and the decoder they are wired into -- computed by an INDEPENDENT
PUBLISHED implementation, HuggingFace ``transformers`` (5.15.0 in this image) ``models/janus/modeling_janus.py``:

    JanusVQVAEResnetBlock   GroupNorm(32, eps 1e-6) -> x*sigmoid(x) -> conv3x3 -> GroupNorm -> swish -> conv3x3, 1x1 ``nin_shortcut`` when in != out
    JanusVQVAEAttnBlock     GroupNorm -> 1x1 q / k / v -> softmax(q k^T * C^-0.5) v -> 1x1 proj_out -> + residual   (one head of width C)
    JanusVQVAEConvUpsample  nearest 2x -> conv3x3
    JanusVQVAEMidBlock      resnet, attention, resnet
    JanusVQVAEDecoder       conv_in -> mid -> per level (num_res_blocks + 1) resnets [+ upsample] -> GroupNorm -> swish -> conv_out

That file is the CompVis latent-diffusion decoder (taming-transformers ``Decoder``), which is what diffusers' ``AutoencoderKL`` decoder -- the
reference's VAE, tld/diffusion.py:91, tld/configs.py:39-43 -- derives from block for block; diffusers itself is absent from this image and
from /opt/wheelhouse.  ``oracle/vae_ref.py`` restates the diffusers graph; this script loads the SAME synthetic diffusers-keyed weights
(``synth_vae_state_dict``) into the Janus modules through the key map below and records inputs -> outputs, so that every block of the
restatement (``_resnet``, ``_attention``, upsample, mid block) and the level wiring are pinned by code the build did not write:

* ``blocks``: each block class alone, constructed exactly as published (equal and unequal channel counts for the resnet);
* ``dec:*``: the whole ``JanusVQVAEDecoder`` with per-stage outputs taken by forward hooks.  ONE deviation from the published constructor, stated
  here and in the fixture: Janus puts an attention block after every resnet of its lowest-resolution level (``attn.append`` when
  ``i_level == num_resolutions - 1``); AutoencoderKL's up blocks have none, so that ModuleList is emptied after construction
  (``len(self.up[i_level].attn) > 0`` in its forward then skips it).  Everything else -- order of levels, ``num_res_blocks + 1``, where the
  upsamplers sit, the final norm / swish / conv -- runs as published.
* what stays "restated from the published graph" only: ``post_quant_conv`` (a 1x1 convolution applied here with ``F.conv2d``) and the
  diffusers KEY NAMES of a real checkpoint.

    python oracle/gen_golden_vae_blocks.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from transformers.models.janus import modeling_janus as mj                                # noqa: E402
from transformers.models.janus.configuration_janus import JanusVQVAEConfig                # noqa: E402

from transformer_latent_diffusion_amd.vae import VaeDecoderConfig, synth_vae_state_dict   # noqa: E402


def _put(module, **tensors):
    sd = module.state_dict()
    assert set(sd) == set(tensors), (sorted(sd), sorted(tensors))
    module.load_state_dict({k: torch.as_tensor(np.asarray(v)).reshape(sd[k].shape) for k, v in tensors.items()})
    return module.eval()


def resnet_tensors(sd, p):
    """diffusers ``ResnetBlock2D`` keys under prefix p -> JanusVQVAEResnetBlock's (conv_shortcut 1x1 == nin_shortcut)."""
    t = {}
    for n in ("norm1", "conv1", "norm2", "conv2"):
        t[n + ".weight"], t[n + ".bias"] = sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"]
    if f"{p}.conv_shortcut.weight" in sd:
        t["nin_shortcut.weight"], t["nin_shortcut.bias"] = sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"]
    return t


def attn_tensors(sd, p):
    """diffusers ``Attention`` (Linear [C, C]) -> JanusVQVAEAttnBlock (1x1 Conv2d [C, C, 1, 1]; load reshapes)."""
    t = {"norm.weight": sd[f"{p}.group_norm.weight"], "norm.bias": sd[f"{p}.group_norm.bias"]}
    for a, b in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "proj_out")):
        t[b + ".weight"], t[b + ".bias"] = sd[f"{p}.{a}.weight"], sd[f"{p}.{a}.bias"]
    return t


def janus_decoder(cfg: VaeDecoderConfig, sd):
    boc = list(cfg.block_out_channels)
    base = boc[0]
    assert all(c % base == 0 for c in boc)
    jc = JanusVQVAEConfig(latent_channels=cfg.latent_channels, base_channels=base, channel_multiplier=[c // base for c in boc],
                          num_res_blocks=cfg.layers_per_block, out_channels=cfg.out_channels, dropout=0.0)
    dec = mj.JanusVQVAEDecoder(jc)
    dec.up[0].attn = torch.nn.ModuleList()            # the one stated deviation: no attention inside AutoencoderKL's up blocks
    t = {"conv_in.weight": sd["decoder.conv_in.weight"], "conv_in.bias": sd["decoder.conv_in.bias"],
         "norm_out.weight": sd["decoder.conv_norm_out.weight"], "norm_out.bias": sd["decoder.conv_norm_out.bias"],
         "conv_out.weight": sd["decoder.conv_out.weight"], "conv_out.bias": sd["decoder.conv_out.bias"]}
    for k, v in resnet_tensors(sd, "decoder.mid_block.resnets.0").items():
        t["mid.block_1." + k] = v
    for k, v in attn_tensors(sd, "decoder.mid_block.attentions.0").items():
        t["mid.attn_1." + k] = v
    for k, v in resnet_tensors(sd, "decoder.mid_block.resnets.1").items():
        t["mid.block_2." + k] = v
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block + 1):
            for k, v in resnet_tensors(sd, f"decoder.up_blocks.{i}.resnets.{j}").items():
                t[f"up.{i}.block.{j}.{k}"] = v
        if i != len(boc) - 1:
            t[f"up.{i}.upsample.conv.weight"] = sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"]
            t[f"up.{i}.upsample.conv.bias"] = sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"]
    return _put(dec, **t), jc


def decode_with_stages(cfg, sd, z):
    """(image, [(stage name in oracle/vae_ref.py's vocabulary, tensor)]) from the Janus decoder."""
    dec, _ = janus_decoder(cfg, sd)
    stages = []
    hook = lambda name: (lambda _m, _i, o: stages.append((name, o.detach().clone())))
    dec.conv_in.register_forward_hook(hook("conv_in"))
    dec.mid.block_1.register_forward_hook(hook("mid.res0"))
    dec.mid.attn_1.register_forward_hook(hook("mid.attn"))
    dec.mid.block_2.register_forward_hook(hook("mid.res1"))
    for i in range(len(cfg.block_out_channels)):
        for j in range(cfg.layers_per_block + 1):
            dec.up[i].block[j].register_forward_hook(hook(f"up{i}.res{j}"))
        if i != len(cfg.block_out_channels) - 1:
            dec.up[i].upsample.register_forward_hook(hook(f"up{i}.upsample"))
    dec.conv_out.register_forward_pre_hook(lambda _m, i: stages.append(("norm_out", i[0].detach().clone())))   # after norm_out + in-place swish
    with torch.no_grad():
        x = F.conv2d(z, torch.as_tensor(sd["post_quant_conv.weight"]), torch.as_tensor(sd["post_quant_conv.bias"]))
        img = dec(x)
    return img, stages


def main():
    out = {"transformers_version": np.array(__import__("transformers").__version__)}
    # ---- 1. the block classes alone, as published ---------------------------------------------------------------------------------------------
    cfg = VaeDecoderConfig(block_out_channels=(64, 128), layers_per_block=1)
    seed = 21
    sd = synth_vae_state_dict(cfg, seed)
    out["blocks_boc"], out["blocks_layers"], out["blocks_seed"] = np.array(cfg.block_out_channels), np.array(cfg.layers_per_block), np.array(seed)
    jc = JanusVQVAEConfig(dropout=0.0)
    g = torch.Generator().manual_seed(31)
    with torch.no_grad():
        x128 = torch.randn(2, 128, 8, 8, generator=g) * 1.4
        x128b = torch.randn(2, 128, 12, 12, generator=g) * 0.9
        res_eq = _put(mj.JanusVQVAEResnetBlock(jc, 128, 128), **resnet_tensors(sd, "decoder.mid_block.resnets.0"))
        res_ne = _put(mj.JanusVQVAEResnetBlock(jc, 128, 64), **resnet_tensors(sd, "decoder.up_blocks.1.resnets.0"))
        attn = _put(mj.JanusVQVAEAttnBlock(128), **attn_tensors(sd, "decoder.mid_block.attentions.0"))
        ups = _put(mj.JanusVQVAEConvUpsample(128), **{"conv.weight": sd["decoder.up_blocks.0.upsamplers.0.conv.weight"],
                                                      "conv.bias": sd["decoder.up_blocks.0.upsamplers.0.conv.bias"]})
        mid_t = {}
        for k, v in resnet_tensors(sd, "decoder.mid_block.resnets.0").items():
            mid_t["block_1." + k] = v
        for k, v in attn_tensors(sd, "decoder.mid_block.attentions.0").items():
            mid_t["attn_1." + k] = v
        for k, v in resnet_tensors(sd, "decoder.mid_block.resnets.1").items():
            mid_t["block_2." + k] = v
        mid = _put(mj.JanusVQVAEMidBlock(jc, 128), **mid_t)
        # (the Janus forwards work in place on their argument's normalised copy only; inputs are cloned anyway)
        out["blk:x128"], out["blk:x128b"] = x128.numpy(), x128b.numpy()
        out["blk:resnet_equal"] = res_eq(x128.clone()).numpy()                 # decoder.mid_block.resnets.0
        out["blk:resnet_unequal"] = res_ne(x128b.clone()).numpy()              # decoder.up_blocks.1.resnets.0 (128 -> 64, 1x1 shortcut)
        out["blk:attention"] = attn(x128.clone()).numpy()                      # decoder.mid_block.attentions.0
        out["blk:attention_b"] = attn(x128b.clone()).numpy()                   # 144 tokens
        out["blk:upsample"] = ups(x128.clone()).numpy()                        # decoder.up_blocks.0.upsamplers.0
        out["blk:mid"] = mid(x128b.clone()).numpy()                            # decoder.mid_block
    # ---- 2. the wired decoder, tiny geometry, every stage ---------------------------------------------------------------------------------------
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(32)) * 1.5
    img, stages = decode_with_stages(cfg, sd, z)
    out["dec:z"], out["dec:image"] = z.numpy(), img.numpy()
    for n, t in stages:
        out["dec:stage:" + n] = t.numpy()
    # ---- 3. the wired decoder at the SDXL-VAE geometry ((128, 256, 512, 512), 2 layers per block: 49.5 M parameters), 8 x 8 latents -> 64 x 64 px;
    # ----    stages as (mean, rms) + a fixed strided sample, the image in full ---------------------------------------------------------------------
    cfg2, seed2 = VaeDecoderConfig(), 22
    sd2 = synth_vae_state_dict(cfg2, seed2)
    z2 = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(33)) * 1.2
    img2, stages2 = decode_with_stages(cfg2, sd2, z2)
    out["sdxl:seed"], out["sdxl:z"], out["sdxl:image"] = np.array(seed2), z2.numpy(), img2.numpy()
    out["sdxl:stage_names"] = np.array([n for n, _ in stages2])
    for n, t in stages2:
        f = t.reshape(-1)
        out["sdxl:stat:" + n] = np.array([float(f.mean()), float(f.pow(2).mean().sqrt())])
        out["sdxl:sample:" + n] = f[::max(1, f.numel() // 2048)][:2048].numpy()
    path = os.path.join(REPO, "tests", "golden", "g18_vae_janus.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
    for k, v in out.items():
        if k.startswith(("blk:", "dec:image", "sdxl:image")):
            print(f"  {k:22s} {v.shape} rms {float(np.sqrt((v.astype(np.float64) ** 2).mean())):.3f}")


if __name__ == "__main__":
    main()
