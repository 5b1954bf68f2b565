"""Byte-level BPE tokenizer producing CLIP token ids — own implementation of the published CLIP BPE scheme, validated
against ids captured from the reference's ``SimpleTokenizer`` (models/clip/simple_tokenizer.py:62-132) and
``clip.tokenize`` (models/clip/clip.py:185-221); see tests/test_tokenizer.py.

The merge table ``bpe_simple_vocab_16e6.txt.gz`` is third-party data (OpenAI CLIP) that is NOT shipped here; it is looked up
at run time: $LPI_BPE_VOCAB, then ./models/clip/ (the reference checkout the plugin is dropped into), then next to this
file.  Callers that already hold token ids (bench, tests) never need it.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache

import regex

VOCAB_FILE = "bpe_simple_vocab_16e6.txt.gz"
SOT_TEXT, EOT_TEXT = "<|startoftext|>", "<|endoftext|>"
N_MERGES = 49152 - 256 - 2      # merges used by CLIP (simple_tokenizer.py:66)
_WORD_END = "</w>"
_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
_WS = regex.compile(r"\s+")


def find_vocab(path: str | None = None) -> str:
    cands = [path, os.environ.get("LPI_BPE_VOCAB"), os.path.join(os.getcwd(), "models", "clip", VOCAB_FILE),
             os.path.join(os.path.dirname(os.path.abspath(__file__)), VOCAB_FILE)]
    for c in cands:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError(f"{VOCAB_FILE} not found; set LPI_BPE_VOCAB or pass token ids instead of strings")


@lru_cache()
def byte_alphabet():
    """256 printable stand-ins, one per byte: printable latin-1 bytes map to themselves, the rest to U+0100.. in order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
    for b in range(256):
        if b not in keep:
            table[b] = chr(256 + extra)
            extra += 1
    return table


try:
    import ftfy as _ftfy
except ImportError:          # ftfy only repairs mojibake; plain text is unchanged by it
    _ftfy = None


def _clean(text: str) -> str:
    if _ftfy is not None:
        text = _ftfy.fix_text(text)
    if "&" in text:
        text = html.unescape(html.unescape(text))
    return _WS.sub(" ", text.strip()).strip().lower()


class SimpleTokenizer:
    def __init__(self, bpe_path: str | None = None):
        lines = gzip.open(find_vocab(bpe_path)).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:N_MERGES + 1]]
        alpha = byte_alphabet()
        # vocabulary order of CLIP: 256 byte symbols in the order printable-first (as bytes_to_unicode enumerates them),
        # the same with the word-end marker, one entry per merge, then the two specials
        ordered = [alpha[b] for b in list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))]
        ordered += [alpha[b] for b in range(256) if alpha[b] not in set(ordered)]
        symbols = ordered + [s + _WORD_END for s in ordered] + ["".join(m) for m in merges] + [SOT_TEXT, EOT_TEXT]
        self.encoder = {s: i for i, s in enumerate(symbols)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.alpha = alpha
        self._memo = {}
        self._text_memo = {}       # whole cleaned string -> ids: dataset captions repeat every epoch (the hot loop is host-bound otherwise)

    def _merge_word(self, word: str):
        """Greedy lowest-rank-first pair merging of one pre-token (already mapped to the byte alphabet)."""
        got = self._memo.get(word)
        if got is not None:
            return got
        parts = list(word[:-1]) + [word[-1] + _WORD_END]
        while len(parts) > 1:
            best, where = None, None
            for i in range(len(parts) - 1):
                r = self.rank.get((parts[i], parts[i + 1]))
                if r is not None and (best is None or r < best):
                    best, where = r, (parts[i], parts[i + 1])
            if best is None:
                break
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == where[0] and parts[i + 1] == where[1]:
                    merged.append(parts[i] + parts[i + 1])
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        ids = [self.encoder[p] for p in parts]
        self._memo[word] = ids
        return ids

    def encode(self, text: str):
        got = self._text_memo.get(text)
        if got is not None:
            return got
        out = []
        alpha = self.alpha
        for tok in _SPLIT.findall(_clean(text)):
            if tok in (SOT_TEXT, EOT_TEXT):
                out.append(self.encoder[tok])
                continue
            ids = self._memo.get(tok)
            if ids is None:
                ids = self._merge_word("".join(alpha[b] for b in tok.encode("utf-8")))
                self._memo[tok] = ids
            out.extend(ids)
        if len(self._text_memo) < 2_000_000:
            self._text_memo[text] = out
        return out


class NativeTokenizer:
    """The same tokenizer with the split / merge / lookup done in C++ (``lpi_bpe_*`` in liblpi_hip.so, csrc/bpe.hip); text cleaning
    stays here (Python's Unicode machinery).  Same ``encode`` / ``encoder`` surface as SimpleTokenizer for the ids this path needs."""

    def __init__(self, bpe_path: str | None = None):
        import ctypes
        from lpi_amd import _lib
        self._lib = _lib.load()
        raw = gzip.open(find_vocab(bpe_path)).read()
        self._h = self._lib.lpi_bpe_create(raw, len(raw))
        if not self._h:
            raise ValueError("malformed BPE merge table")
        self._ctypes = ctypes
        self._buf = (ctypes.c_int32 * 4096)()
        self.encoder = {SOT_TEXT: 49406, EOT_TEXT: 49407}
        ids = self.encode("x")
        self.encoder["x</w>"] = ids[0]

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.lpi_bpe_destroy(h)

    def encode(self, text: str):
        n = self._lib.lpi_bpe_encode(self._h, _clean(text).encode("utf-8"), self._buf, len(self._buf))
        if n < 0:
            raise ValueError(f"lpi_bpe_encode failed with code {n}")
        if n > len(self._buf):
            self._buf = (self._ctypes.c_int32 * (2 * n))()
            return self.encode(text)
        return list(self._buf[:n])

    def tokenize(self, texts, context_length: int = 77, truncate: bool = False):
        """clip.tokenize in one native call: int64 [n, context_length] (numpy)."""
        import numpy as np
        ct = self._ctypes
        enc = [_clean(t).encode("utf-8") for t in texts]
        arr = (ct.c_char_p * len(enc))(*enc)
        out = np.zeros((len(enc), context_length), dtype=np.int64)
        rc = self._lib.lpi_bpe_tokenize(self._h, arr, len(enc), context_length, int(truncate), out.ctypes.data_as(ct.c_void_p))
        if rc > 0:
            raise RuntimeError(f"Input {texts[rc - 1]} is too long for context length {context_length}")
        if rc < 0:
            raise ValueError(f"lpi_bpe_tokenize failed with code {rc}")
        return out


def tokenize(tokenizer, texts, context_length: int = 77, truncate: bool = False):
    """[SOT] + ids + [EOT], zero padded to context_length; RuntimeError when too long (clip.py:205-219)."""
    import torch
    if isinstance(texts, str):
        texts = [texts]
    if isinstance(tokenizer, NativeTokenizer):
        return torch.from_numpy(tokenizer.tokenize(list(texts), context_length, truncate))
    import numpy as np
    sot, eot = tokenizer.encoder[SOT_TEXT], tokenizer.encoder[EOT_TEXT]
    out = np.zeros((len(texts), context_length), dtype=np.int64)
    for i, t in enumerate(texts):
        ids = tokenizer.encode(t)
        n = len(ids) + 2
        if n > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length - 2]
            n = context_length
        out[i, 0] = sot
        out[i, 1:n - 1] = ids
        out[i, n - 1] = eot
    return torch.from_numpy(out)
