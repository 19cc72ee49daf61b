"""CLIP's text tokeniser on the host: ``clip.tokenize(prompts, truncate=True)`` without the ``clip`` package.

The reference turns a prompt into the label the denoiser sees with (tld/diffusion.py:136-140)

    text_tokens = clip.tokenize(label, truncate=True).to(device)
    model.encode_text(text_tokens)

``clip`` (openai/CLIP, installed from git, unpinned) is a third-party dependency that is neither in the reference checkout nor in this
image.  This module restates its published algorithm (clip/simple_tokenizer.py ``SimpleTokenizer`` + clip/clip.py ``tokenize``):
lower-cased, whitespace-collapsed text is split by CLIP's regular expression, every piece goes byte -> printable-unicode symbol, byte-pair
merges are applied in rank order with the end-of-word marker ``</w>``, and the ids are framed by <|startoftext|> / <|endoftext|> and
zero-padded (or truncated, keeping the end token) to ``context_length``.

The vocabulary comes from the merges file the CLIP repository ships (``bpe_simple_vocab_16e6.txt.gz``, first line a header, then
48 894 merges); it cannot be fetched here, so the caller passes its path -- or a merges list.  **Parity**: pinned in the CPU suite
(tests/test_clip_tokenizer.py) against HuggingFace ``transformers.CLIPTokenizer`` (the Rust ``tokenizers`` BPE -- an independent published
implementation of the same tokeniser) on a synthetic vocabulary built the way CLIP builds its own; with the real merges file the two agree on
everything ``ftfy`` would leave untouched (``ftfy.fix_text`` is applied when importable, exactly as openai/CLIP does).
"""
from __future__ import annotations

import gzip
import html
import unicodedata
from functools import lru_cache
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import re as _stdre
import torch

SOT, EOT = "<|startoftext|>", "<|endoftext|>"
_PATTERN = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""
N_MERGES = 49152 - 256 - 2          # CLIP's vocabulary: 256 byte symbols, the same with </w>, 48 894 merges, two specials


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """Every byte as one printable unicode character (the GPT-2 table): the printable latin-1 bytes map to themselves, the other 68
    to code points 256, 257, ... in byte order."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def byte_symbols() -> List[str]:
    """The 256 byte symbols in CLIP's vocabulary order: the self-mapped bytes first, then the remapped ones."""
    t = bytes_to_unicode()
    keep = [b for b in range(256) if t[b] == chr(b)]
    rest = [b for b in range(256) if t[b] != chr(b)]
    return [t[b] for b in keep + rest]


def build_vocab(merges: Sequence[Tuple[str, str]]) -> Dict[str, int]:
    sym = byte_symbols()
    vocab = sym + [s + "</w>" for s in sym] + ["".join(m) for m in merges] + [SOT, EOT]
    return {tok: i for i, tok in enumerate(vocab)}


def read_merges(bpe_path: str) -> List[Tuple[str, str]]:
    """``bpe_simple_vocab_16e6.txt[.gz]``: a header line, then one merge per line; CLIP uses the first 48 894."""
    opener = gzip.open if str(bpe_path).endswith(".gz") else open
    with opener(bpe_path, "rb") as f:
        lines = f.read().decode("utf-8").split("\n")
    return [tuple(ln.split()) for ln in lines[1:N_MERGES + 1] if ln.strip()]


def _clean(text: str) -> str:
    try:                                     # openai/CLIP's basic_clean starts with ftfy.fix_text (mojibake repair, NFC, ...)
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:                      # not in this image: the NFC step alone, which is also what HuggingFace's normaliser does
        text = unicodedata.normalize("NFC", text)
    text = html.unescape(html.unescape(text)).strip()
    return _stdre.sub(r"\s+", " ", text).strip()


class ClipTokenizer:
    def __init__(self, bpe_path: Optional[str] = None, merges: Optional[Iterable[Tuple[str, str]]] = None):
        if (bpe_path is None) == (merges is None):
            raise ValueError("pass either the path of CLIP's merges file (bpe_simple_vocab_16e6.txt.gz) or a list of merges")
        self.merges = read_merges(bpe_path) if bpe_path is not None else [tuple(m) for m in merges]
        self.encoder = build_vocab(self.merges)
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(self.merges)}
        self._byte = bytes_to_unicode()
        try:                     # CLIP's pattern uses \p{L} / \p{N}: the third-party ``regex`` module, needed by tokeniser users only
            import regex
        except ImportError as e:  # pragma: no cover -- present in this image
            raise ImportError("ClipTokenizer needs the 'regex' package (unicode property classes in CLIP's split pattern); "
                              "pip install regex.  Inference from text embeddings and training do not need it.") from e
        self._pat = regex.compile(_PATTERN, regex.IGNORECASE)
        self._cache: Dict[str, Tuple[str, ...]] = {SOT: (SOT,), EOT: (EOT,)}
        self.sot, self.eot = self.encoder[SOT], self.encoder[EOT]

    def _bpe(self, piece: str) -> Tuple[str, ...]:
        """Merge the symbols of one pre-token: repeatedly the adjacent pair of lowest rank, all of its occurrences left to right."""
        hit = self._cache.get(piece)
        if hit is not None:
            return hit
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word, word[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            out, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    out.append(best[0] + best[1]); i += 2
                else:
                    out.append(word[i]); i += 1
            word = out
        res = tuple(word)
        self._cache[piece] = res
        return res

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for piece in self._pat.findall(_clean(text).lower()):
            sym = "".join(self._byte[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(sym))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        inv = {c: b for b, c in self._byte.items()}
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(inv[c] for c in text if c in inv).decode("utf-8", errors="replace").replace("</w>", " ")

    def tokenize(self, texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """clip.tokenize: [n, context_length] int64 (torch >= 1.8 in openai/CLIP: int32 there, LongTensor before; the text towers index an
        embedding with it either way), <|startoftext|> ids <|endoftext|> zero-padded; too long: truncated with the end token kept, or an error."""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, text in enumerate(texts):
            toks = [self.sot] + self.encode(text) + [self.eot]
            if len(toks) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {text} is too long for context length {context_length}")
                toks = toks[:context_length]
                toks[-1] = self.eot
            out[i, :len(toks)] = torch.tensor(toks, dtype=torch.long)
        return out
