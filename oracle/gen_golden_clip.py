"""Writes tests/golden/g13_clip_text.npz: the CLIP text tower pinned against an INDEPENDENT published implementation.

The reference calls ``model.encode_text(tokens)`` of openai/CLIP "ViT-L/14" (tld/diffusion.py:136-140,160-161).  The ``clip`` package is
absent here, but HuggingFace ``transformers`` (5.15.0 in this image) ships the same text tower as ``CLIPTextModelWithProjection``
(``hidden_act="quick_gelu"``, causal mask, EOS pooling, bias-free projection).  This script, run in the BUILD container only:

  1. builds the deterministic synthetic weights of ``synth_clip_state_dict(cfg, seed)`` (openai/CLIP key names -- what the engine and
     ``oracle/clip_ref.py`` load; 495 MB for ViT-L/14, so the fixture stores the seed and a checksum, not the tensors),
  2. maps them onto the HF module's ``state_dict`` (q/k/v split of ``attn.in_proj_*``, ``text_projection`` transposed into the
     ``nn.Linear`` layout) and loads them with ``strict=True``,
  3. runs ``clip.tokenize``-shaped ids (SOT, words, EOT = the largest id, zero padding) through the HF model and stores
     tokens -> ``text_embeds`` (= ``encode_text``'s output) and ``last_hidden_state`` (after ``ln_final``).

Two geometries: ViT-L/14 (12 layers, width 768, 12 heads, 77 positions, projection 768) and a tiny one for the CPU suite.
EOS pooling: HF takes the first position whose id equals ``eos_token_id`` (or, for the legacy ``eos_token_id == 2``, the arg-max id);
openai/CLIP takes ``text.argmax(-1)``.  With ``eos_token_id`` = vocab_size - 1 = CLIP's EOT id both are the same position for
tokenizer-shaped ids, which is what the fixture uses (and asserts).

    python oracle/gen_golden_clip.py
"""
import os
import sys
import zlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from transformer_latent_diffusion_amd.clip_text import ClipTextConfig, synth_clip_state_dict   # noqa: E402


def to_hf_state_dict(cfg: ClipTextConfig, sd):
    """openai/CLIP text-side keys -> transformers.CLIPTextModelWithProjection keys."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    w = cfg.width
    out = {
        "text_model.embeddings.token_embedding.weight": t(sd["token_embedding.weight"]),
        "text_model.embeddings.position_embedding.weight": t(sd["positional_embedding"]),
        "text_model.final_layer_norm.weight": t(sd["ln_final.weight"]),
        "text_model.final_layer_norm.bias": t(sd["ln_final.bias"]),
        "text_projection.weight": t(sd["text_projection"].T),          # x @ text_projection == Linear(weight = text_projection^T)
    }
    for i in range(cfg.layers):
        s, d = f"transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
        wi, bi = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):            # nn.MultiheadAttention packs [q; k; v]
            out[d + f"self_attn.{n}.weight"] = t(wi[j * w:(j + 1) * w])
            out[d + f"self_attn.{n}.bias"] = t(bi[j * w:(j + 1) * w])
        out[d + "self_attn.out_proj.weight"] = t(sd[s + "attn.out_proj.weight"])
        out[d + "self_attn.out_proj.bias"] = t(sd[s + "attn.out_proj.bias"])
        out[d + "layer_norm1.weight"] = t(sd[s + "ln_1.weight"]); out[d + "layer_norm1.bias"] = t(sd[s + "ln_1.bias"])
        out[d + "layer_norm2.weight"] = t(sd[s + "ln_2.weight"]); out[d + "layer_norm2.bias"] = t(sd[s + "ln_2.bias"])
        out[d + "mlp.fc1.weight"] = t(sd[s + "mlp.c_fc.weight"]); out[d + "mlp.fc1.bias"] = t(sd[s + "mlp.c_fc.bias"])
        out[d + "mlp.fc2.weight"] = t(sd[s + "mlp.c_proj.weight"]); out[d + "mlp.fc2.bias"] = t(sd[s + "mlp.c_proj.bias"])
    return out


def tokens(cfg: ClipTextConfig, batch: int, seed: int) -> torch.Tensor:
    """clip.tokenize-shaped ids: SOT (vocab - 2), words, EOT (vocab - 1 = the largest id), zero padding; one full-length row."""
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(batch, cfg.context_length, dtype=torch.long)
    for b in range(batch):
        n = cfg.context_length - 2 if b == 0 else int(torch.randint(1, cfg.context_length - 2, (1,), generator=g))
        t[b, 0] = cfg.vocab_size - 2
        t[b, 1:1 + n] = torch.randint(1, cfg.vocab_size - 2, (n,), generator=g)
        t[b, 1 + n] = cfg.vocab_size - 1
    return t


def weights_crc(sd) -> int:
    c = 0
    for k in sd:
        c = zlib.crc32(np.ascontiguousarray(sd[k]).tobytes(), c)
    return c


def run_hf(cfg: ClipTextConfig, sd, text: torch.Tensor):
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    hf_cfg = CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.width, intermediate_size=4 * cfg.width,
                            projection_dim=cfg.embed_dim, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                            max_position_embeddings=cfg.context_length, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                            attention_dropout=0.0, pad_token_id=0, bos_token_id=cfg.vocab_size - 2, eos_token_id=cfg.vocab_size - 1)
    try:
        hf_cfg._attn_implementation = "eager"
    except Exception:
        pass
    model = CLIPTextModelWithProjection(hf_cfg).eval().to(torch.float32)
    hf_sd = to_hf_state_dict(cfg, sd)
    own = {k: v for k, v in model.state_dict().items() if "position_ids" not in k}
    assert set(own) == set(hf_sd), (sorted(set(own) ^ set(hf_sd))[:6])
    for k in own:
        assert tuple(own[k].shape) == tuple(hf_sd[k].shape), k
    model.load_state_dict(hf_sd, strict=False)              # (strict up to the non-parameter position_ids buffer, asserted above)
    with torch.no_grad():
        out = model(input_ids=text)
    return out.text_embeds.float(), out.last_hidden_state.float()


def main():
    torch.manual_seed(0)
    out = {}
    for tag, cfg, seed, batch in (("tiny", ClipTextConfig(vocab_size=1000, context_length=16, width=128, heads=2, layers=2, embed_dim=64), 7, 5),
                                  ("l14", ClipTextConfig(), 0, 4)):
        sd = synth_clip_state_dict(cfg, seed)
        text = tokens(cfg, batch, 100 + seed)
        assert (text.argmax(-1) == (text == cfg.vocab_size - 1).int().argmax(-1)).all()       # the two EOS pooling rules coincide
        emb, hid = run_hf(cfg, sd, text)
        out[f"{tag}:cfg"] = np.array([cfg.vocab_size, cfg.context_length, cfg.width, cfg.heads, cfg.layers, cfg.embed_dim])
        out[f"{tag}:seed"] = np.array(seed)
        out[f"{tag}:weights_crc32"] = np.array(weights_crc(sd), dtype=np.uint32)
        out[f"{tag}:tokens"] = text.numpy().astype(np.int32)
        out[f"{tag}:text_embeds"] = emb.numpy()
        out[f"{tag}:last_hidden_state"] = hid.numpy()[:2]                                        # (two samples keep the file small)
        # cross-check on the spot: the repo's restatement against the HF module
        from oracle.clip_ref import TorchRefClipText
        mine = TorchRefClipText(cfg, sd).encode_text(text)
        rel = float((mine - emb).pow(2).mean().sqrt() / emb.pow(2).mean().sqrt())
        print(f"{tag}: B={batch} embeds rms {float(emb.pow(2).mean().sqrt()):.4f}  oracle/clip_ref vs transformers rel-rms {rel:.2e}")
    import transformers
    out["transformers_version"] = np.array(transformers.__version__)
    path = os.path.join(REPO, "tests", "golden", "g13_clip_text.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
