"""CLIP BPE tokeniser (clip_tokenizer.py) pinned against HuggingFace's CLIPTokenizer (the Rust `tokenizers` BPE): an independent
published implementation of the tokeniser behind the reference's ``clip.tokenize(label, truncate=True)`` (tld/diffusion.py:136).
No vocabulary file exists offline, so both are built from the same synthetic merges, learned here by plain BPE over a small corpus
and laid out the way CLIP lays out its vocabulary (256 byte symbols, the same with </w>, merges, <|startoftext|>, <|endoftext|>)."""
import collections
import gzip

import pytest
import torch

from transformer_latent_diffusion_amd.clip_tokenizer import (ClipTokenizer, build_vocab, byte_symbols, bytes_to_unicode)

CORPUS = ("a photo of a cat sitting on the mat . the quick brown fox jumps over the lazy dog ! it's a painting of mountains , "
          "rivers and 12 trees in the style of monet's water lilies ; photographs photographer painted painter they're we've i'm "
          "you'll he'd don't café naïve 2024 100% (oil on canvas) high-resolution 8k").split()

PROMPTS = ["a photo of a cat", "The  Quick brown FOX!!", "it's monet's   painting, 1234 trees...", "café naïve — ünïcode 日本語",
           "  leading and trailing  ", "", "photographers' paintings: mountains&rivers (2024)", "they're WE'VE i'm You'll he'd don't",
           "tabs\tand\nnewlines\r\n here", "&amp;lt;b&amp;gt; html &quot;entities&quot;", "emoji \U0001F600 and symbols ©®™ §¶", "decomposed cafe\u0301 nai\u0308ve",
           "a " * 100, "x" * 300, "<|startoftext|> inside <|endoftext|> text"]


def learn_merges(words, n):
    b2u = bytes_to_unicode()
    table = collections.Counter()
    for w in words:
        sym = [b2u[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        table[tuple(sym)] += 1
    merges = []
    for _ in range(n):
        pairs = collections.Counter()
        for w, c in table.items():
            for pr in zip(w, w[1:]):
                pairs[pr] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda pr: pairs[pr])
        merges.append(best)
        nxt = collections.Counter()
        for w, c in table.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            nxt[tuple(out)] += c
        table = nxt
    return merges


@pytest.fixture(scope="module")
def merges():
    m = learn_merges(CORPUS, 160)
    assert len(m) > 100
    return m


def test_vocabulary_layout(merges):
    sym = byte_symbols()
    assert len(sym) == len(set(sym)) == 256 and sym[0] == "!" and sym[ord("a") - ord("!")] == "a"
    v = build_vocab(merges)
    assert len(v) == 512 + len(merges) + 2
    assert v["!"] == 0 and v["!</w>"] == 256 and v["".join(merges[0])] == 512
    assert v["<|startoftext|>"] == len(v) - 2 and v["<|endoftext|>"] == len(v) - 1
    # with CLIP's 48 894 merges this puts the specials at 49406 / 49407, the ids the text towers are trained with
    assert 512 + (49152 - 256 - 2) == 49406


def test_matches_huggingface_clip_tokenizer(merges):
    transformers = pytest.importorskip("transformers")
    hf = transformers.CLIPTokenizer(vocab=build_vocab(merges), merges=[tuple(m) for m in merges])
    mine = ClipTokenizer(merges=merges)
    for text in PROMPTS:
        if "<|" in text or "&" in text and ";" in text:
            continue                      # markers / html entities: SimpleTokenizer-only behaviour, covered in test_tokenize_contract
        got = mine.tokenize(text, truncate=True)[0].tolist()
        want = hf(text, padding="max_length", max_length=77, truncation=True)["input_ids"]
        n = got.index(mine.eot) + 1
        assert got[:n] == want[:n], text
        assert all(t == 0 for t in got[n:])            # clip.tokenize pads with zeros (HF pads with its pad token)
        assert got[0] == mine.sot and len(got) == 77


def test_tokenize_contract(merges):
    tok = ClipTokenizer(merges=merges)
    out = tok.tokenize(["a cat", "the dog !"], truncate=True)
    assert out.shape == (2, 77) and out.dtype == torch.long
    assert tok.tokenize("a cat").tolist() == out[:1].tolist()
    long = "a " * 100
    with pytest.raises(RuntimeError, match="too long"):
        tok.tokenize(long)
    cut = tok.tokenize(long, truncate=True)[0]
    assert cut[0] == tok.sot and cut[-1] == tok.eot and (cut != 0).all()
    ids = tok.encode("<|startoftext|> inside <|endoftext|> text")
    assert ids[0] == tok.sot and tok.eot in ids        # the markers are single tokens inside text, as in SimpleTokenizer
    assert tok.decode(tok.encode("The quick brown fox's 12 trees!")) == "the quick brown fox 's 1 2 trees ! "
    assert tok.encode("&amp;lt;b&amp;gt; html &quot;entities&quot;") == tok.encode('<b> html "entities"')   # basic_clean unescapes twice
    assert tok.encode("cafe\u0301") == tok.encode("caf\u00e9")
    assert tok.encode("A  CAT") == tok.encode("a cat") == tok.encode(" a\tcat\n")


def test_reads_clip_merges_file(tmp_path, merges):
    path = tmp_path / "bpe_simple_vocab.txt.gz"
    with gzip.open(path, "wb") as f:
        f.write(('"bpe_simple_vocab_16e6.txt#version: 0.2\n' + "\n".join(" ".join(m) for m in merges) + "\n").encode("utf-8"))
    a, b = ClipTokenizer(bpe_path=str(path)), ClipTokenizer(merges=merges)
    assert a.merges == b.merges
    for text in PROMPTS:
        assert torch.equal(a.tokenize(text, truncate=True), b.tokenize(text, truncate=True))
    with pytest.raises(ValueError):
        ClipTokenizer()


def test_pipeline_tokenises_prompts_without_clip_package(merges):
    """DiffusionTransformer(tokenizer=...) feeds clip_model.encode_text the ids clip.tokenize(prompts, truncate=True) would."""
    from transformer_latent_diffusion_amd.diffusion import DiffusionTransformer
    tok = ClipTokenizer(merges=merges)
    shell = DiffusionTransformer.__new__(DiffusionTransformer)
    shell._tokenizer, shell._text_encoder, shell.device = tok, None, torch.device("cpu")
    seen = {}

    class Tower:
        def encode_text(self, ids):
            seen["ids"] = ids
            return torch.zeros(ids.shape[0], 768)

    shell.clip_model = Tower()
    out = shell.encode_text(["a cat", "a " * 100])
    assert out.shape == (2, 768) and torch.equal(seen["ids"], tok.tokenize(["a cat", "a " * 100], truncate=True))
