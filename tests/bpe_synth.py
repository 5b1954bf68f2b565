"""A SYNTHETIC byte-level BPE merge table in the format of CLIP's ``bpe_simple_vocab_16e6.txt.gz`` (a header line, then one merge "a b" per line):
49 152 - 256 - 2 valid merges drawn by a seeded generator, every operand an existing symbol.  It holds nothing of the reference's data — the ids it
yields are NOT CLIP's — but any valid table exercises the whole tokenizer path (clean -> split -> merge by rank -> lookup -> [SOT] ids [EOT] padding),
so the native (C++) and the Python tokenizer can be compared on a box that has no copy of the real table (the GPU box)."""
import gzip
import random

from lpi_amd.retrieval.models.clip import simple_tokenizer as T


def write_table(path, seed=0):
    rng = random.Random(seed)
    ab = T.byte_alphabet()
    base = [ab[b] for b in range(256)]
    lower = [ab[ord(c)] for c in "abcdefghijklmnopqrstuvwxyz"]
    common = lower + [ab[ord(c)] for c in "0123456789'.,-!?"] + [ab[b] for b in (0xC3, 0xA9, 0xE4, 0xBD, 0xA0, 0xF0, 0x9F, 0x98, 0x80)]
    open_syms = list(base)            # symbols that do not end a word: may stand on the left of a merge
    open_hot = list(lower)            # ... those made of common characters (so that the merges apply to ordinary text)
    closed = [s + "</w>" for s in base]
    closed_hot = [s + "</w>" for s in common]
    have = set(open_syms) | set(closed)
    merges = []
    while len(merges) < T.N_MERGES:
        hot = rng.random() < 0.85
        a = rng.choice(open_hot if hot else open_syms)
        right_closed = rng.random() < 0.4
        b = rng.choice((closed_hot if hot else closed) if right_closed else (open_hot if hot else open_syms))
        new = a + b
        if new in have or len(new) > 24:
            continue
        have.add(new)
        merges.append((a, b))
        if right_closed:
            closed.append(new)
            if hot:
                closed_hot.append(new)
        else:
            open_syms.append(new)
            if hot:
                open_hot.append(new)
    text = '"synthetic merge table#version: 0.2\n' + "\n".join(f"{a} {b}" for a, b in merges) + "\n"
    with gzip.open(path, "wb") as f:
        f.write(text.encode("utf-8"))
    return str(path)


def fuzz_cases(seed=7, n=300):
    rng = random.Random(seed)
    alphabet = ("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 .,;:!?'\"-_()[]{}<>|/\\@#$%^&*+=~`\t\n"
                "éèüñçøßÆŒ¿¡€£¥©®°±²³½¾×÷ "
                "αβγδЖдёשלוםمرحبا你好世界こんにちは한국어๑๒٣४５ⅣⅫ😀🎉👍🏽✈️‍")
    words = ["don't", "it's", "we're", "I've", "I'm", "they'll", "he'd", "'sx", "'tis", "a's", "<|startoftext|>", "<|endoftext|>",
             "&amp;", "&lt;b&gt;", "naïve", "co-op", "3.14", "x2y", "...", "a photo of a dog", "two people riding bicycles down the street"]
    cases = list(words)
    for _ in range(n):
        k = rng.randint(0, 40)
        s = "".join(rng.choice(alphabet) for _ in range(k))
        if rng.random() < 0.5:
            s = " ".join([s, rng.choice(words), rng.choice(words)])
        cases.append(s)
    return cases
