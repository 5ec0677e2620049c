"""Synthetic IPADIC-shaped dictionary and Japanese-like corpora.

The real mecab-ipadic data is absent from the reference checkout
(.MISSING_LARGE_BLOBS) and cannot be fetched, so every measurement and parity
run uses this generator.  Shapes follow public IPADIC 2.7.0 documentation:
392 000 lexicon records, a 1316x1316 i16 connection matrix, the 11 char.def
categories with their invoke/group flags, 40 unk.def rows.  Deterministic for a
given seed (numpy PCG64).  Record ids follow the reference builder's order:
records sorted by (surface bytes, left_id, right_id, cost) and numbered from 1
(kanpyo-dict/src/builder/record.rs:5-19, builder.rs:49-53); unknown rows sorted
by category name and numbered from 1 (unk_dict.rs:19-42).
"""
from __future__ import annotations

from dataclasses import dataclass

import os

import numpy as np

from .dict import Dict

SEED_DICT = 0x4B414E50

# (name, invoke, group) in char.def declaration order => category byte value
CATEGORIES = [
    ("DEFAULT", 0, 1), ("SPACE", 0, 1), ("KANJI", 0, 0), ("SYMBOL", 1, 1), ("NUMERIC", 1, 1), ("ALPHA", 1, 1),
    ("HIRAGANA", 0, 1), ("KATAKANA", 1, 1), ("KANJINUMERIC", 1, 1), ("GREEK", 1, 1), ("CYRILLIC", 1, 1),
]
CAT = {n: i for i, (n, _, _) in enumerate(CATEGORIES)}
# unk.def rows per category (sum 40)
UNK_ROWS = {"DEFAULT": 1, "SPACE": 1, "KANJI": 6, "SYMBOL": 1, "NUMERIC": 1, "ALPHA": 6, "HIRAGANA": 6,
            "KATAKANA": 6, "KANJINUMERIC": 1, "GREEK": 6, "CYRILLIC": 5}
N_CONTEXT = 1316

_RANGES = {
    "SPACE": [(0x20, 0x20), (0x09, 0x0D)],
    "SYMBOL": [(0x21, 0x2F), (0x3A, 0x40), (0x5B, 0x60), (0x7B, 0x7E), (0xA1, 0xBF), (0x2000, 0x206F),
               (0x20A0, 0x20CF), (0x2100, 0x214F), (0x2190, 0x22FF), (0x2460, 0x257F), (0x25A0, 0x26FE),
               (0x3000, 0x303F), (0xFF01, 0xFF0F), (0xFF1A, 0xFF20), (0xFF3B, 0xFF40), (0xFF5B, 0xFF65),
               (0xFFE0, 0xFFEF)],
    "NUMERIC": [(0x30, 0x39), (0xFF10, 0xFF19)],
    "ALPHA": [(0x41, 0x5A), (0x61, 0x7A), (0xC0, 0x236), (0x1E00, 0x1EF9), (0xFF21, 0xFF3A), (0xFF41, 0xFF5A)],
    "HIRAGANA": [(0x3041, 0x309F)],
    "KATAKANA": [(0x30A1, 0x30FF), (0x31F0, 0x31FF), (0xFF66, 0xFF9F)],
    "KANJI": [(0x2E80, 0x2EF3), (0x2F00, 0x2FD5), (0x3005, 0x3005), (0x3007, 0x3007), (0x3400, 0x4DB5),
              (0x4E00, 0x9FA5), (0xF900, 0xFA2D), (0xFA30, 0xFA6A)],
    "GREEK": [(0x374, 0x3FB)],
    "CYRILLIC": [(0x400, 0x4F9), (0x500, 0x50F)],
}
_KANJINUMERIC = [0x4E00, 0x4E8C, 0x4E09, 0x56DB, 0x4E94, 0x516D, 0x4E03, 0x516B, 0x4E5D, 0x5341, 0x767E, 0x5343,
                 0x4E07, 0x5104, 0x5146]


def char_category_table() -> np.ndarray:
    t = np.zeros(1 << 16, dtype=np.uint8)  # builder/char_def.rs:33: 65 536 entries, default class 0
    for name in ("SPACE", "SYMBOL", "NUMERIC", "ALPHA", "HIRAGANA", "KATAKANA", "KANJI", "GREEK", "CYRILLIC"):
        for lo, hi in _RANGES[name]:
            t[lo : hi + 1] = CAT[name]
    for cp in _KANJINUMERIC:
        t[cp] = CAT["KANJINUMERIC"]
    return t


def _zipf_weights(n: int, s: float, shift: float = 2.0) -> np.ndarray:
    w = 1.0 / np.power(np.arange(n) + shift, s)
    return w / w.sum()


@dataclass
class SynthDict:
    dict: Dict
    surfaces: list  # unique surfaces (str), index aligned with `weights`
    weights: np.ndarray  # sampling weights for corpus generation
    n_records: int


def _make_words(rng, pool, weights, count, len_lo, len_hi, len_p, seen, out):
    """Draw `count` new unique words from a weighted char pool."""
    made = 0
    while made < count:
        m = max(1024, (count - made) * 2)
        lens = np.clip(rng.geometric(len_p, size=m) + (len_lo - 1), len_lo, len_hi)
        chars = rng.choice(pool, size=int(lens.sum()), p=weights)
        pos = 0
        for ln in lens:
            w = "".join(map(chr, chars[pos : pos + ln]))
            pos += ln
            if w not in seen:
                seen.add(w)
                out.append(w)
                made += 1
                if made >= count:
                    break


def build_dict(n_records: int = 392_000, seed: int = SEED_DICT, n_context: int = N_CONTEXT, dense: bool = False) -> SynthDict:
    """The synthetic IPADIC-shaped dictionary (deterministic in its arguments).  KANPYO_SYNTH_CACHE=<dir>: the built object is kept there as a pickle
    and loaded by later processes (the measurement tools set it: 20 s of building per process otherwise); unset = always built."""
    cache = os.environ.get("KANPYO_SYNTH_CACHE")
    if not cache:
        return _build_dict(n_records, seed, n_context, dense)
    import pickle
    path = os.path.join(cache, f"synth_{n_records}_{seed}_{n_context}_{int(dense)}.pkl")
    try:
        with open(path, "rb") as fh:
            return pickle.load(fh)
    except (OSError, pickle.UnpicklingError, EOFError, AttributeError):
        pass
    sd = _build_dict(n_records, seed, n_context, dense)
    try:
        os.makedirs(cache, exist_ok=True)
        tmp = path + f".{os.getpid()}"
        with open(tmp, "wb") as fh:
            pickle.dump(sd, fh, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(tmp, path)
    except OSError:
        pass
    return sd


def _build_dict(n_records: int, seed: int, n_context: int, dense: bool) -> SynthDict:
    """dense=True: the same record count laid out for the lattice density real IPADIC text shows (N ~ 8-10 x C; the default shape
    gives 5.4): more short hiragana / kanji surfaces and nested prefixes (more dictionary words per start position), more records per surface
    (homographs: up to 12 on the single-kana particles), corpus weights that favour them -- buckets of more than eight predecessors at about
    half of the positions (the default: 13 %), which is where the sweep's P <= 16 / P <= 32 bodies and the MAXM = 8 parked-prefix limit are met."""
    rng = np.random.default_rng(seed)
    hira = np.arange(0x3041, 0x3094)
    kata = np.arange(0x30A1, 0x30F7)
    kanji = rng.permutation(np.arange(0x4E00, 0x9FA6))
    kanji = kanji[~np.isin(kanji, _KANJINUMERIC)]
    w_hira = _zipf_weights(hira.size, 0.9)
    w_kata = _zipf_weights(kata.size, 0.7)
    w_kanji = _zipf_weights(kanji.size, 1.05, shift=20.0)
    alpha = np.array(list(range(0x41, 0x5B)) + list(range(0x61, 0x7B)))
    w_alpha = _zipf_weights(alpha.size, 0.5)

    seen: set = set()
    words: list = []
    for c in hira:  # every single hiragana is a word (particles, auxiliaries)
        words.append(chr(c)); seen.add(chr(c))
    n_single_hira = len(words)
    tgt_unique = int(n_records / (1.9 if dense else 1.178))  # ~15 % of surfaces carry 1-8 extra records (dense: about half of them)
    plan = [
        (hira, w_hira, 0.070, 2, 7, 0.50), (kanji, w_kanji, 0.020, 1, 1, 0.99), (kanji, w_kanji, 0.520, 2, 12, 0.62),
        (kata, w_kata, 0.080, 2, 12, 0.38), (alpha, w_alpha, 0.004, 2, 10, 0.35),
    ] if not dense else [
        (hira, w_hira, 0.160, 2, 6, 0.55), (kanji, w_kanji, 0.030, 1, 1, 0.99), (kanji, w_kanji, 0.420, 2, 8, 0.75),
        (kata, w_kata, 0.060, 2, 10, 0.40), (alpha, w_alpha, 0.004, 2, 10, 0.35),
    ]
    for pool, w, frac, lo, hi, p in plan:
        _make_words(rng, pool, w, int(tgt_unique * frac), lo, hi, p, seen, words)
    # kanji stem + hiragana ending (verbs / adjectives)
    n_mixed = int(tgt_unique * 0.10)
    made = 0
    while made < n_mixed:
        k = rng.integers(1, 3)
        h = rng.integers(1, 4)
        w = "".join(map(chr, rng.choice(kanji, size=k, p=w_kanji))) + "".join(map(chr, rng.choice(hira, size=h, p=w_hira)))
        if w not in seen:
            seen.add(w); words.append(w); made += 1
    # forced nested prefixes: the rest extend an earlier surface by 1-3 chars
    while len(words) < tgt_unique:
        base = words[int(rng.integers(0, len(words)))]
        if len(base) > 10:
            continue
        last = ord(base[-1])
        pool, w = (hira, w_hira) if 0x3041 <= last <= 0x3093 else (kata, w_kata) if 0x30A1 <= last <= 0x30F6 else (kanji, w_kanji)
        ext = base + "".join(map(chr, rng.choice(pool, size=int(rng.integers(1, 4)), p=w)))
        if ext not in seen:
            seen.add(ext); words.append(ext)

    # records per surface
    nrec = np.ones(len(words), dtype=np.int64)
    multi = rng.random(len(words)) < (0.50 if dense else 0.15)
    nrec[multi] += np.minimum(rng.geometric(0.40 if dense else 0.45, size=int(multi.sum())), 7)
    nrec[:n_single_hira] = rng.integers(5, 13, size=n_single_hira) if dense else rng.integers(3, 9, size=n_single_hira)
    diff = n_records - int(nrec.sum())
    i = n_single_hira
    idle = 0
    while diff != 0:  # hit the record count exactly
        if diff > 0 and nrec[i] < (12 if dense else 8):
            nrec[i] += 1; diff -= 1; idle = 0
        elif diff < 0 and nrec[i] > 1:
            nrec[i] -= 1; diff += 1; idle = 0
        else:
            idle += 1
            if idle > len(words):
                raise ValueError(f"build_dict: {n_records} records cannot be met with {len(words)} surfaces (use >= 5000 records)")
        i = i + 1 if i + 1 < len(words) else n_single_hira
    total = int(nrec.sum())
    surf_of = np.repeat(np.arange(len(words)), nrec)
    ctx_w = _zipf_weights(n_context - 1, 1.1, shift=3.0)
    left = rng.choice(np.arange(1, n_context), size=total, p=ctx_w)
    right = np.where(rng.random(total) < 0.9, left, rng.choice(np.arange(1, n_context), size=total, p=ctx_w))
    wlen = np.array([len(w) for w in words])[surf_of]
    cost = np.clip(rng.normal(3500 + 1100 * np.minimum(wlen, 6), 2500, size=total), -2000, 15000).astype(np.int64)
    single = surf_of < n_single_hira
    cost[single] = rng.integers(200, 4000, size=int(single.sum()))
    enc = [w.encode("utf-8") for w in words]
    order = sorted(range(total), key=lambda r: (enc[surf_of[r]], left[r], right[r], cost[r]))
    order = np.array(order)
    sorted_keywords = [enc[surf_of[r]] for r in order]
    morphs = np.stack([left[order], right[order], cost[order]], axis=1)

    matrix = rng.integers(-3000, 3001, size=(n_context, n_context))  # [left][right] == flat left*rows+right
    matrix[0, :] = rng.integers(-500, 501, size=n_context)
    matrix[:, 0] = rng.integers(-500, 501, size=n_context)

    unk_map, unk_morphs, nxt = {}, [], 1
    for name in sorted(UNK_ROWS):  # reference sorts unk.def rows by category string first
        cnt = UNK_ROWS[name]
        unk_map[CAT[name]] = (nxt, cnt)
        for _ in range(cnt):
            c = int(rng.integers(1, n_context))
            unk_morphs.append((c, c, int(rng.integers(3000, 12000))))
        nxt += cnt

    d = Dict.from_parts(
        sorted_keywords, morphs, n_context, n_context, matrix.reshape(-1), [c[0] for c in CATEGORIES],
        char_category_table(), np.array([c[1] for c in CATEGORIES], dtype=np.uint8),
        np.array([c[2] for c in CATEGORIES], dtype=np.uint8), unk_map, unk_morphs,
    )
    # corpus sampling weights: short / early words are frequent
    wts = 1.0 / (np.arange(len(words)) % 5003 + 5.0) / np.maximum(np.array([len(w) for w in words]), 1) ** 1.5
    wts[:n_single_hira] *= 40.0
    is_hira = np.array([0x3041 <= ord(w[0]) <= 0x3093 and 0x3041 <= ord(w[-1]) <= 0x3093 for w in words])
    wts[is_hira] *= 4.0
    if dense:  # the corpus leans on what makes lattices dense: short surfaces with many records and many longer words starting with them
        wts *= np.asarray(nrec, dtype=np.float64) ** 0.5
    wts /= wts.sum()
    return SynthDict(d, words, wts, total)


# ------------------------------------------------------------------- corpora

def _noise_segment(rng, kind: int, heavy: bool) -> str:
    if kind == 0:  # katakana run (loanword not in the lexicon)
        n = int(rng.integers(2, 41 if heavy else 9))
        return "".join(map(chr, rng.integers(0x30A1, 0x30F7, size=n)))
    if kind == 1:  # digits
        return "".join(map(chr, rng.integers(0x30, 0x3A, size=int(rng.integers(1, 9)))))
    if kind == 2:  # ASCII alnum
        pool = np.array(list(range(0x41, 0x5B)) + list(range(0x61, 0x7B)) + list(range(0x30, 0x3A)))
        return "".join(map(chr, rng.choice(pool, size=int(rng.integers(2, 13)))))
    if kind == 3:  # isolated symbols / punctuation
        return chr(int(rng.choice([0x3001, 0x3002, 0x300C, 0x300D, 0x30FB, 0xFF01, 0xFF1F, 0x2026, 0x20, 0x25CB])))
    if kind == 4:  # rare kanji (mostly not in the lexicon)
        return "".join(map(chr, rng.integers(0x4E00, 0x9FA6, size=int(rng.integers(1, 4)))))
    if kind == 5:  # non-BMP: category falls back to table[0] (char_category_def.rs:37)
        return "".join(map(chr, rng.integers(0x20000, 0x2A6D7, size=int(rng.integers(1, 4)))))
    if kind == 6:  # greek / cyrillic
        base = 0x391 if rng.random() < 0.5 else 0x410
        return "".join(map(chr, rng.integers(base, base + 24, size=int(rng.integers(2, 8)))))
    return chr(int(rng.integers(0x3041, 0x3094)))


def make_corpus(sd: SynthDict, n: int, seed: int, kind: str = "cfg2") -> list:
    """kind: 'cfg2' (~40 chars, N(40,8) clamp [8,96]), 'cfg3' (log-uniform 8..512, unknown-heavy),
    'cfg5' (2048 chars with a >1024 same-category run)."""
    rng = np.random.default_rng(seed)
    if kind == "cfg2":
        mean = float(os.environ.get("KANPYO_CFG2_MEAN", "40"))  # (tools only: the same shape at another mean length -- LDS-footprint experiments)
        lens = np.clip(np.rint(rng.normal(mean, mean / 5, size=n)), 8, 96).astype(int)
        p_noise, heavy, kinds = 0.15, False, [0, 1, 2, 3, 4, 6]
    elif kind == "cfg3":
        lens = np.exp(rng.uniform(np.log(8), np.log(512), size=n)).astype(int)
        p_noise, heavy, kinds = 0.22, True, [0, 0, 1, 2, 3, 4, 5, 6]
    elif kind == "cfg5":
        lens = np.full(n, 2048, dtype=int)
        p_noise, heavy, kinds = 0.10, True, [0, 1, 2, 3, 4, 5, 6]
    else:
        raise ValueError(kind)
    avg_word = 2.2
    pool = rng.choice(len(sd.surfaces), size=int(lens.sum() / avg_word * 1.3) + 1024, p=sd.weights)
    pi = 0
    out = []
    for L in lens:
        parts, have = [], 0
        if kind == "cfg5":  # one groupable run longer than the 1024 cap (lattice.rs:80-82)
            run_len = int(rng.integers(1100, 1400))
            run_at = int(rng.integers(0, L - run_len))
            run_cp = int(rng.integers(0x30A1, 0x30F7)) if rng.random() < 0.5 else int(rng.integers(0x30, 0x3A))
        while have < L:
            if kind == "cfg5" and have >= run_at and run_at >= 0:
                parts.append("".join(map(chr, rng.integers(0x30A1, 0x30F7, size=run_len))) if run_cp >= 0x30A1
                             else "".join(map(chr, rng.integers(0x30, 0x3A, size=run_len))))
                have += run_len
                run_at = -1
                continue
            if rng.random() < p_noise / 2.5:  # noise segments average ~2.5x a word's length
                seg = _noise_segment(rng, int(rng.choice(kinds)), heavy)
            else:
                if pi >= pool.size:
                    pool = rng.choice(len(sd.surfaces), size=pool.size, p=sd.weights); pi = 0
                seg = sd.surfaces[pool[pi]]; pi += 1
            parts.append(seg)
            have += len(seg)
        out.append("".join(parts)[:L])
    return out


def dense_case(rng):
    """A tiny alphabet with many duplicate records per surface (rng: random.Random): buckets of 1..200 predecessors, up to
    ~150 targets per position -- every shape of the sweep step (P <= 8, <= 16, <= 32, beyond; T beyond one pass; T > 127)
    and the parked-match overflow (more than 8 prefixes at a position), which the IPADIC-shaped corpora never reach.
    -> (Dict, sentences)"""
    nr = np.random.default_rng(rng.randrange(1 << 30))
    alpha = "あいうえおか"[: rng.choice([2, 3, 6])]
    words = set()
    for _ in range(rng.choice([20, 80, 300])):
        words.add("".join(nr.choice(list(alpha), size=int(nr.integers(1, rng.choice([3, 5, 12]))))))
    recs = []
    for w in sorted(words, key=lambda x: x.encode()):
        recs += [w] * int(nr.choice([1, 1, 2, 3, 9, rng.choice([17, 33, 70])]))
    nctx = rng.choice([1, 3, 40])
    morphs = np.stack([nr.integers(0, nctx, len(recs)), nr.integers(0, nctx, len(recs)), nr.integers(-2000, 9000, len(recs))], axis=1)
    cat = np.zeros(65536, dtype=np.uint8)
    for ch in alpha:
        cat[ord(ch)] = 1
    unk = {0: (1, 1), 1: (2, rng.choice([1, 3]))}
    um = [(0, 0, 5000)] + [(int(nr.integers(0, nctx)), int(nr.integers(0, nctx)), int(nr.integers(1000, 9000))) for _ in range(unk[1][1])]
    d = Dict.from_parts(recs, morphs, nctx, nctx, nr.integers(-3000, 3000, nctx * nctx), ["DEFAULT", "H"], cat,
                        np.array([0, rng.choice([0, 1])], dtype=np.uint8), np.array([1, rng.choice([0, 1])], dtype=np.uint8), unk, um)
    sents = ["".join(nr.choice(list(alpha + "xy"), size=int(nr.integers(1, rng.choice([8, 40, 120])))))
             for _ in range(rng.choice([1, 7, 128, 129, 700, 3000]))]
    return d, sents


def width_case(rng):
    """Like dense_case, over an alphabet of 1-, 2-, 3- and 4-byte characters (and U+FFFF): keys that are prefixes of each other across widths,
    characters beyond the BMP at any place of a key -- what the character-level copy of the trie (kgpu_chartrie.cpp) has to get right.
    -> (Dict, sentences)"""
    nr = np.random.default_rng(rng.randrange(1 << 30))
    pool = list("abé\u00dfλЖあい漢字\uffff") + ["\U00020BB7", "\U0001F600", "\U00029E3D", "\U0001D4B3"]
    alpha = rng.sample(pool, rng.choice([3, 6, len(pool)]))
    words = set()
    for _ in range(rng.choice([10, 60, 400])):
        words.add("".join(nr.choice(alpha, size=int(nr.integers(1, rng.choice([3, 5, 9]))))))
    recs = []
    for w in sorted(words, key=lambda x: x.encode()):
        recs += [w] * int(nr.choice([1, 1, 2, 5]))
    nctx = rng.choice([1, 4, 30])
    morphs = np.stack([nr.integers(0, nctx, len(recs)), nr.integers(0, nctx, len(recs)), nr.integers(-2000, 9000, len(recs))], axis=1)
    cat = np.zeros(65536, dtype=np.uint8)
    for ch in alpha:
        if ord(ch) < 65536 and rng.random() < 0.7:
            cat[ord(ch)] = 1
    unk = {0: (1, 1), 1: (2, rng.choice([1, 2]))}
    um = [(0, 0, 5000)] + [(int(nr.integers(0, nctx)), int(nr.integers(0, nctx)), int(nr.integers(1000, 9000))) for _ in range(unk[1][1])]
    d = Dict.from_parts(recs, morphs, nctx, nctx, nr.integers(-3000, 3000, nctx * nctx), ["DEFAULT", "H"], cat,
                        np.array([0, rng.choice([0, 1])], dtype=np.uint8), np.array([1, rng.choice([0, 1])], dtype=np.uint8), unk, um)
    text = alpha + list("xy\U0001F601")
    sents = ["".join(nr.choice(text, size=int(nr.integers(1, rng.choice([8, 40, 200, 1500])))))
             for _ in range(rng.choice([1, 7, 128, 129, 700, 3000]))]
    return d, sents


EDGE_SENTENCES = ["", "あ", "ア" * 700, "a" * 300, "𠮷野家で𩸽", "すもももももももものうち", "　　", "1234567890" * 40, "。" * 65]


def mixed_case(sd: SynthDict, rng, sizes=(1, 5, 50, 120, 700, 4096, 9000)):
    """One shuffled batch over the IPADIC-shaped generator: cfg 2 text, often cfg 3, sometimes cfg 5 documents, edge sentences."""
    mix = make_corpus(sd, rng.choice(list(sizes)), rng.randrange(1 << 30), "cfg2")
    if rng.random() < 0.7:
        mix += make_corpus(sd, rng.choice([3, 100, 600]), rng.randrange(1 << 30), "cfg3")
    if rng.random() < 0.3:
        mix += make_corpus(sd, rng.choice([1, 4]), rng.randrange(1 << 30), "cfg5")
    mix += rng.sample(EDGE_SENTENCES, rng.randrange(len(EDGE_SENTENCES)))
    rng.shuffle(mix)
    return mix
