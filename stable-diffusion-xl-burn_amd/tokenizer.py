"""Byte-level BPE tokenizers of the Embedder: host-side string processing, mirrors of the reference's
`ClipTokenizer` (src/token/clip.rs:79-230) and `OpenClipTokenizer` (src/token/open_clip.rs:70-221).

The two share vocabulary and merges; they differ in where those are read from, in the special-token cache (the OpenCLIP
variant has none, open_clip.rs:94-99) and in the padding id (49407 vs 0, clip.rs:227-229 / open_clip.rs:218-220).
The asset files are NOT part of this repository (they belong to the reference / OpenAI CLIP): pass `tokenizer_dir` or set
SDXL_TOKENIZER_DIR to a directory laid out like the reference's `tokenizer/` (clip/bpe_simple_vocab_16e6.txt,
open_clip/{merges,vocab}.txt).  A Rust caller keeps using the reference's own tokenizer -- only tensors cross the C ABI.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

SOT, EOT = 49406, 49407
_PAT = (r"(?i)<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|\p{L}+|\p{N}|[^\s\p{L}\p{N}]+")   # clip.rs:110


def default_tokenizer_dir() -> str:
    for c in (os.environ.get("SDXL_TOKENIZER_DIR"), "tokenizer", "/root/reference/tokenizer"):
        if c and os.path.isdir(c):
            return c
    raise FileNotFoundError("tokenizer assets not found: set SDXL_TOKENIZER_DIR to a directory with "
                            "clip/bpe_simple_vocab_16e6.txt and open_clip/{merges,vocab}.txt (the reference's tokenizer/)")


def bytes_to_unicode() -> List[Tuple[int, str]]:
    """clip.rs:11-31: printable latin-1 bytes map to themselves, the rest to U+0100.. in byte order"""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = [chr(b) for b in bs]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(chr(256 + n))
            n += 1
    return list(zip(bs, cs))


def _load_merges(path: str) -> List[Tuple[str, str]]:
    """clip.rs:43-61: every line with at least two whitespace-separated words contributes its first two"""
    merges = []
    with open(path, encoding="utf-8") as fh:
        for line in fh:
            w = line.split()
            if len(w) >= 2:
                merges.append((w[0], w[1]))
    return merges


class _BpeTokenizer:
    def __init__(self, vocab: List[str], merges: List[Tuple[str, str]], cache: Dict[str, str], pad: int):
        import regex   # \p{L} / \p{N} classes, as the reference's `regex` crate
        bu = bytes_to_unicode()
        self.byte_encoder = {b: c for b, c in bu}
        self.byte_decoder = {c: b for b, c in bu}
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.cache = dict(cache)
        self.pat = regex.compile(_PAT)
        self._pad = pad

    def bpe(self, token: str) -> str:
        """clip.rs:123-178"""
        if token in self.cache:
            return self.cache[token]
        word = list(token)
        if word:
            word[-1] += "</w>"
        pairs = list(zip(word, word[1:]))
        if not pairs:
            return token + "</w>"
        while True:
            ranked = [p for p in pairs if p in self.bpe_ranks]
            if not ranked:
                break
            first, second = min(ranked, key=lambda p: self.bpe_ranks[p])
            new_word, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new_word.extend(word[i:])
                    break
                new_word.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new_word.append(first + second)
                    i += 2
                else:
                    new_word.append(word[i])
                    i += 1
            word = new_word
            if len(word) == 1:
                break
            pairs = list(zip(word, word[1:]))
        return " ".join(word)

    def encode(self, text: str, add_sot: bool, add_eot: bool) -> List[int]:
        """Tokenizer::encode (token/mod.rs:5, clip.rs:182-210)"""
        cleaned = " ".join(text.strip().split()).lower()
        out: List[int] = [SOT] if add_sot else []
        for m in self.pat.finditer(cleaned):
            token = "".join(self.byte_encoder[b] for b in m.group(0).encode("utf-8"))
            out.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        if add_eot:
            out.append(EOT)
        return out

    def decode(self, tokens: List[int]) -> str:
        text = "".join(self.decoder[t] for t in tokens)
        return bytes(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def start_of_text_token(self) -> int:
        return SOT

    def end_of_text_token(self) -> int:
        return EOT

    def padding_token(self) -> int:
        return self._pad


class ClipTokenizer(_BpeTokenizer):
    """reference ClipTokenizer::new (clip.rs:91-121): merges = lines 1 .. 49152-256-2 of the OpenAI BPE file, vocabulary
    built from the byte alphabet, pad = end-of-text"""

    def __init__(self, tokenizer_dir: Optional[str] = None):
        d = tokenizer_dir or default_tokenizer_dir()
        merges = _load_merges(os.path.join(d, "clip", "bpe_simple_vocab_16e6.txt"))[1:49152 - 256 - 2 + 1]
        chars = [c for _, c in bytes_to_unicode()]
        vocab = chars + [c + "</w>" for c in chars] + [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
        cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        super().__init__(vocab, merges, cache, EOT)


class OpenClipTokenizer(_BpeTokenizer):
    """reference OpenClipTokenizer::new (open_clip.rs:82-113): vocabulary and merges read from files, no special-token
    cache, pad = 0"""

    def __init__(self, tokenizer_dir: Optional[str] = None):
        d = tokenizer_dir or default_tokenizer_dir()
        merges = _load_merges(os.path.join(d, "open_clip", "merges.txt"))
        with open(os.path.join(d, "open_clip", "vocab.txt"), encoding="utf-8") as fh:
            vocab = [line.rstrip("\n") for line in fh]
        super().__init__(vocab, merges, {}, 0)


def tokenize_text(text: str, tokenizer: _BpeTokenizer, seq_len: int) -> List[int]:
    """stablediffusion/mod.rs:785-801: encode with sot/eot, pad (or truncate) to seq_len with the tokenizer's pad id"""
    ids = tokenizer.encode(text, True, True)
    ids = ids[:seq_len] + [tokenizer.padding_token()] * max(0, seq_len - len(ids))
    return ids
