"""A SYNTHETIC byte-level BPE merge table in the format of CLIP's ``bpe_simple_vocab_16e6.txt.gz`` (a header line, then one merge "a b" per line):
49 152 - 256 - 2 valid merges drawn by a seeded generator, every operand an existing symbol.  It holds nothing of the reference's data — the ids it
yields are NOT CLIP's — but any valid table exercises the whole tokenizer path (clean -> split -> merge by rank -> lookup -> [SOT] ids [EOT] padding),
so the native (C++) and the Python tokenizer can be compared on a box that has no copy of the real table (the GPU box), and bench.py's plugin_step
record can feed caption STRINGS through the plugin's own tokenizer path there (``ensure_vocab``)."""
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


def ensure_vocab(cache_dir=None, seed=0) -> str:
    """Path of a merge table for the tokenizer: CLIP's own when one is found (simple_tokenizer.find_vocab), else the synthetic table, written once per
    cache directory (default: $TMPDIR) and announced through $LPI_BPE_VOCAB so that every later lookup of this process finds it."""
    import os
    import tempfile
    try:
        return T.find_vocab()
    except FileNotFoundError:
        pass
    d = cache_dir or tempfile.gettempdir()
    path = os.path.join(d, f"lpi_synthetic_bpe_{seed}.txt.gz")
    if not os.path.isfile(path):
        tmp = f"{path}.{os.getpid()}.tmp"
        write_table(tmp, seed)
        os.replace(tmp, path)
    os.environ["LPI_BPE_VOCAB"] = path
    return path


_WORDS = ("a an the of on in at with and two three man woman dog cat bus train plate pizza kite bench street table field water grass snow beach sign "
          "red blue white black green large small young old sitting standing riding holding flying parked looking eating playing next front top near "
          "people person group kitchen room park city road tree bird horse sheep cow boat clock phone laptop bed couch cake wine cup bowl fork").split()


def captions(n: int, seed: int = 0, min_chars: int = 5, max_chars: int = 40):
    """n synthetic caption strings of COCO-like words whose letter count is uniform in [min_chars, max_chars] — with a byte-level BPE table a caption
    costs at most one token per letter, so "X"*16 + caption + "." always fits the 77-token context (clip.py:213-218 raises otherwise), and the token
    counts spread like SURVEY 8(d)'s synthetic ids (5..40 caption tokens)."""
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        budget = rng.randint(min_chars, max_chars)
        words, used = [], 0
        while True:
            w = rng.choice(_WORDS)
            if used + len(w) > budget:
                break
            words.append(w)
            used += len(w)
        out.append(" ".join(words) if words else "a")
    return out


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
