"""Own BPE tokenizer vs ids captured from the reference's SimpleTokenizer / clip.tokenize.  Needs the third-party merge
table (not shipped); skipped where it is absent (e.g. on the GPU box)."""
import os

import numpy as np
import pytest

from lpi_amd.retrieval.models.clip import simple_tokenizer as T

CANDS = [os.environ.get("LPI_BPE_VOCAB"), "/root/reference/retrieval/models/clip/" + T.VOCAB_FILE]
VOCAB = next((c for c in CANDS if c and os.path.isfile(c)), None)
needs_vocab = pytest.mark.skipif(VOCAB is None, reason="bpe_simple_vocab_16e6.txt.gz not available")


@needs_vocab
def test_ids_match_reference(golden):
    tk = T.SimpleTokenizer(VOCAB)
    g = golden("tokenizer")
    ids = T.tokenize(tk, [str(t) for t in g["texts"]]).numpy()
    assert (ids == g["ids"]).all()
    assert tk.encoder[T.SOT_TEXT] == 49406 and tk.encoder[T.EOT_TEXT] == 49407 and tk.encoder["x</w>"] == 343


@needs_vocab
def test_prompt_learner_ids(golden):
    from lpi_amd.retrieval.models.clip import prompt_learner as PL
    PL._tokenizer = T.SimpleTokenizer(VOCAB)
    g = golden("tiny_eval")
    pl = PL.PromptLearner(PL.cfgc())
    assert (pl([str(c) for c in g["captions"]]).numpy() == g["token_ids"]).all()


@needs_vocab
def test_too_long_raises():
    tk = T.SimpleTokenizer(VOCAB)
    with pytest.raises(RuntimeError):
        T.tokenize(tk, ["word " * 100])
    assert T.tokenize(tk, ["word " * 100], truncate=True)[0, -1] == 49407


def test_missing_vocab_is_loud(tmp_path, monkeypatch):
    monkeypatch.delenv("LPI_BPE_VOCAB", raising=False)
    monkeypatch.chdir(tmp_path)
    if os.path.isfile(os.path.join(os.path.dirname(T.__file__), T.VOCAB_FILE)):
        pytest.skip("vocab shipped next to the module")
    with pytest.raises(FileNotFoundError):
        T.find_vocab(None)


def test_synthetic_ids_shape():
    from lpi_amd import synth
    ids = synth.token_ids(5)
    assert ids.shape == (5, 77) and (ids[:, 0] == 49406).all() and (ids[:, 1:17] == 343).all()
    assert (ids.max(1) == 49407).all() and (np.argmax(ids, 1) >= 23).all()


@needs_vocab
def test_native_tokenizer_matches_reference_ids_and_python(golden):
    """The C++ BPE (lpi_bpe_* in liblpi_hip.so) against the ids captured from the reference, and against the Python implementation
    on a fuzz of ASCII / Latin / CJK / emoji / digit / punctuation / contraction mixes (both are this repo's own code; the fixture is
    the reference's output)."""
    import random
    tk_py = T.SimpleTokenizer(VOCAB)
    tk = T.NativeTokenizer(VOCAB)
    g = golden("tokenizer")
    texts = [str(t) for t in g["texts"]]
    assert (T.tokenize(tk, texts).numpy() == g["ids"]).all()
    assert tk.encoder["x</w>"] == 343
    rng = random.Random(7)
    alphabet = ("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 .,;:!?'\"-_()[]{}<>|/\\@#$%^&*+=~`\t\n"
                "éèüñçøßÆŒ¿¡€£¥©®°±²³½¾×÷ "
                "αβγδЖдёשלוםمرحبا你好世界こんにちは한국어๑๒٣४５ⅣⅫ😀🎉👍🏽✈️‍")
    words = ["don't", "it's", "we're", "I've", "I'm", "they'll", "he'd", "'sx", "'tis", "a's", "<|startoftext|>", "<|endoftext|>",
             "&amp;", "&lt;b&gt;", "naïve", "co-op", "3.14", "x2y", "..."]
    cases = list(words)
    for _ in range(400):
        n = rng.randint(0, 40)
        s = "".join(rng.choice(alphabet) for _ in range(n))
        if rng.random() < 0.5:
            s = " ".join([s, rng.choice(words), rng.choice(words)])
        cases.append(s)
    for s in cases:
        assert tk.encode(s) == tk_py.encode(s), repr(s)
    short = [c for c in cases if len(tk_py.encode(c)) <= 75]
    assert (T.tokenize(tk, short).numpy() == T.tokenize(tk_py, short).numpy()).all()
    with pytest.raises(RuntimeError):
        T.tokenize(tk, ["word " * 100])
    assert T.tokenize(tk, ["word " * 100], truncate=True)[0, -1] == 49407


def test_native_tokenizer_matches_python_on_a_synthetic_merge_table(tmp_path):
    """No copy of CLIP's merge table needed: a seeded synthetic table of the same format and size (tests/bpe_synth.py).  The C++ BPE and the
    Python one must agree id for id on the fuzz, on the padded [n, 77] layout, and on the too-long behaviour."""
    import bpe_synth
    path = bpe_synth.write_table(tmp_path / "synthetic_bpe.txt.gz", seed=3)
    tk_py, tk = T.SimpleTokenizer(path), T.NativeTokenizer(path)
    cases = bpe_synth.fuzz_cases()
    merged = 0
    for s in cases:
        a, b = tk.encode(s), tk_py.encode(s)
        assert a == b, repr(s)
        merged += sum(1 for i in a if i >= 512)
    assert merged > 100      # the table's merges really fire on this text (ids >= 512 are merge results)
    short = [c for c in cases if len(tk_py.encode(c)) <= 75]
    ids = T.tokenize(tk, short).numpy()
    assert (ids == T.tokenize(tk_py, short).numpy()).all()
    assert (ids[:, 0] == 49406).all() and (ids.max(1) == 49407).all()
    with pytest.raises(RuntimeError):
        T.tokenize(tk, ["word " * 100])
    assert T.tokenize(tk, ["word " * 100], truncate=True)[0, -1] == 49407


@needs_vocab
@pytest.mark.parametrize("impl", ["python", "native"])
def test_wide_fixture_of_reference_ids(golden, impl):
    """Round 6: 1 200 strings through the imported reference's `SimpleTokenizer.encode` and `clip.tokenize` (tests/golden/gen_golden.py
    tokenizer_wide_case: COCO-like captions, HTML entities, whitespace runs, mixed case, digits, contractions, non-Latin scripts, emoji, special-token
    strings, long words, > 77-token overflows).  Both of this repo's BPEs — the C++ one the plugin uses and the Python one — must give the reference's ids
    for every string, the same padded rows, and raise on exactly the strings the reference raises on (clip.py:213-218)."""
    g = golden("tokenizer_wide")
    tk = T.SimpleTokenizer(VOCAB) if impl == "python" else T.NativeTokenizer(VOCAB)
    texts = [str(t) for t in g["texts"]]
    assert len(texts) >= 1000
    offs, flat = g["encode_offsets"], g["encode_flat"]
    for i, t in enumerate(texts):
        assert tk.encode(t) == flat[offs[i]:offs[i + 1]].tolist(), (i, repr(t))
    fit = [t for t, long_ in zip(texts, g["too_long"]) if not long_]
    assert (T.tokenize(tk, fit).numpy() == g["rows"][~g["too_long"]]).all()
    for i in np.nonzero(g["too_long"])[0]:
        with pytest.raises(RuntimeError):
            T.tokenize(tk, [texts[i]])
        assert (T.tokenize(tk, [texts[i]], truncate=True).numpy()[0] == g["rows"][i]).all()
