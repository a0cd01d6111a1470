"""Token ids -> the string the application consumes.

Host-side, pure Python (microseconds per crop).  Two steps of the reference's recogniser:

* ``tokenizer.decode(ids, skip_special_tokens=True)`` of the character-level BERT-Japanese
  tokenizer: id -> token through ``vocab.txt``, special tokens dropped,
  ``" ".join(tokens).replace(" ##", "")`` (TF/models/bert_japanese/tokenization_bert_japanese.py:250-262);
* ``post_process`` of the ``manga-ocr`` package [RECALL - the package is not available offline]:
  strip ALL whitespace; ``…`` -> ``...``; every run of two or more of ``・`` / ``.`` -> the same
  number of ``.``; then ``jaconv.h2z(text, ascii=True, digit=True)`` (half-width -> full-width for
  ASCII, digits and katakana).  ``jaconv`` is not installed here; ``h2z`` below restates its tables.
"""
from __future__ import annotations

import os
import re
from typing import Iterable, List, Optional, Sequence

SPECIAL_TOKENS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")

# jaconv's conv_table [RECALL, two independent recollections agree; the package is not installed here]: the
# full-width partners of " ' \ ` are the typographic ” ’ ￥ ‘, not the code-point-shifted ＂ ＇ ＼ ｀.
_HALF_ASCII = "abcdefghijklmnopqrstuvwxyz" + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~ "
_FULL_ASCII = "ａｂｃｄｅｆｇｈｉｊｋｌｍｎｏｐｑｒｓｔｕｖｗｘｙｚ" + "ＡＢＣＤＥＦＧＨＩＪＫＬＭＮＯＰＱＲＳＴＵＶＷＸＹＺ" + \
              "！”＃＄％＆’（）＊＋，－．／：；＜＝＞？＠［￥］＾＿‘｛｜｝～　"
_HALF_DIGIT = "0123456789"
_FULL_DIGIT = "０１２３４５６７８９"
_HALF_KANA = "ｧｱｨｲｩｳｪｴｫｵｶｷｸｹｺｻｼｽｾｿﾀﾁｯﾂﾃﾄﾅﾆﾇﾈﾉﾊﾋﾌﾍﾎﾏﾐﾑﾒﾓｬﾔｭﾕｮﾖﾗﾘﾙﾚﾛﾜｦﾝｰ｡｢｣､･ﾞﾟ"
_FULL_KANA = "ァアィイゥウェエォオカキクケコサシスセソタチッツテトナニヌネノハヒフヘホマミムメモャヤュユョヨラリルレロワヲンー。「」、・゛゜"
_VOICED_HALF = "ｶｷｸｹｺｻｼｽｾｿﾀﾁﾂﾃﾄﾊﾋﾌﾍﾎｳ"
_VOICED_FULL = "ガギグゲゴザジズゼゾダヂヅデドバビブベボヴ"
_SEMI_HALF = "ﾊﾋﾌﾍﾎ"
_SEMI_FULL = "パピプペポ"

_H2Z_A = dict(zip(_HALF_ASCII, _FULL_ASCII))
_H2Z_D = dict(zip(_HALF_DIGIT, _FULL_DIGIT))
_H2Z_K = dict(zip(_HALF_KANA, _FULL_KANA))
_H2Z_V = {h + "ﾞ": f for h, f in zip(_VOICED_HALF, _VOICED_FULL)}
_H2Z_S = {h + "ﾟ": f for h, f in zip(_SEMI_HALF, _SEMI_FULL)}


_H2Z_TR = {(True, False, False): str.maketrans(_H2Z_K),
           (True, True, False): str.maketrans({**_H2Z_K, **_H2Z_A}),
           (True, False, True): str.maketrans({**_H2Z_K, **_H2Z_D}),
           (True, True, True): str.maketrans({**_H2Z_K, **_H2Z_A, **_H2Z_D}),
           (False, True, False): str.maketrans(_H2Z_A),
           (False, False, True): str.maketrans(_H2Z_D),
           (False, True, True): str.maketrans({**_H2Z_A, **_H2Z_D}),
           (False, False, False): {}}


def _h2z_tables(text: str, kana: bool = True, ascii: bool = False, digit: bool = False) -> str:
    """The restated tables: voiced / semi-voiced pairs first (longest match), then one character-wise translate."""
    if kana and ("ﾞ" in text or "ﾟ" in text):
        for pair, full in _H2Z_V.items():
            if pair in text:
                text = text.replace(pair, full)
        for pair, full in _H2Z_S.items():
            if pair in text:
                text = text.replace(pair, full)
    return text.translate(_H2Z_TR[(bool(kana), bool(ascii), bool(digit))])


try:        # the reference's recogniser calls jaconv itself: when the package is installed, so do we
    import jaconv as _jaconv
except ImportError:
    _jaconv = None


def h2z(text: str, kana: bool = True, ascii: bool = False, digit: bool = False) -> str:
    """Half-width -> full-width (jaconv.h2z semantics: voiced / semi-voiced marks combine with
    the preceding half-width katakana).  Uses ``jaconv`` when it is importable, else the restated tables."""
    if _jaconv is not None:
        return _jaconv.h2z(text, kana=kana, ascii=ascii, digit=digit)
    return _h2z_tables(text, kana=kana, ascii=ascii, digit=digit)


def post_process(text: str) -> str:
    text = "".join(text.split())
    text = text.replace("…", "...")
    text = re.sub("[・.]{2,}", lambda m: (m.end() - m.start()) * ".", text)
    return h2z(text, ascii=True, digit=True)


class Vocab:
    """id <-> token table of the decoder (one token per line of ``vocab.txt``)."""

    def __init__(self, tokens: Sequence[str]):
        self.tokens = list(tokens)
        self.special_ids = {i for i, t in enumerate(self.tokens) if t in SPECIAL_TOKENS}
        import numpy as np
        self._arr = np.array(self.tokens + ["[UNK]"], dtype=object)       # last slot: ids outside the table
        self._special = np.zeros(len(self.tokens) + 1, dtype=bool)
        self._special[list(self.special_ids)] = True
        # fast path of ids_to_text: post_process removes ALL whitespace, so the decode's `" ".join(...).replace(" ##", "")`
        # reduces to concatenating the tokens with the "##" of every non-first word piece dropped and the whitespace inside
        # tokens removed; both per-token forms are precomputed (checked against the plain composition in the CPU tests)
        nows = ["".join(t.split()) for t in self.tokens] + ["[UNK]"]
        self._first = np.array(nows, dtype=object)
        self._rest = np.array(["".join(t[2:].split()) if t.startswith("##") else n for t, n in zip(self.tokens + ["[UNK]"], nows)], dtype=object)
        self._plain = not any((" ##" in t) or (t != "##" and t.endswith(" ") ) for t in self.tokens)
        # ... and with post_process's character-wise steps applied per token where they commute with concatenation:
        # "…" -> "..." and the half -> full-width translation of everything except '.' and '･' (the dot-run rule must
        # still see the ASCII dot and must not see a '・' that comes from a half-width '･'); tokens with a (semi-)voiced
        # mark combine with their neighbour and send the row down the plain path
        def pre(t):
            t = t.replace("…", "...")
            return "".join(ch if ch in ".･" else _h2z_tables(ch, ascii=True, digit=True) for ch in t)
        self._first_pre = np.array([pre(t) for t in self._first.tolist()], dtype=object)
        self._rest_pre = np.array([pre(t) for t in self._rest.tolist()], dtype=object)
        self._mark = np.array([("ﾞ" in t) or ("ﾟ" in t) for t in self.tokens] + [False], dtype=bool)

    @classmethod
    def from_file(cls, path: str) -> "Vocab":
        with open(path, "r", encoding="utf-8") as f:
            return cls([line.rstrip("\n") for line in f])

    @classmethod
    def synthetic(cls, size: int = 6144) -> "Vocab":
        """Stand-in vocabulary for the synthetic-weight model: the five special tokens, then one
        CJK ideograph per id (so decoded strings are well-defined and whitespace-free)."""
        toks = list(SPECIAL_TOKENS) + [chr(0x4E00 + i) for i in range(size - len(SPECIAL_TOKENS))]
        return cls(toks)

    def __len__(self) -> int:
        return len(self.tokens)

    def decode(self, ids: Iterable[int], skip_special_tokens: bool = True) -> str:
        import numpy as np
        a = np.asarray(list(ids) if not hasattr(ids, "__len__") else ids, dtype=np.int64).ravel()
        a = np.where((a >= 0) & (a < len(self.tokens)), a, len(self.tokens))
        if skip_special_tokens:
            a = a[~self._special[a]]
        return " ".join(self._arr[a].tolist()).replace(" ##", "").strip()


def ids_to_text(vocab: Vocab, ids: Iterable[int]) -> str:
    """tokenizer.decode(ids, skip_special_tokens=True) followed by post_process."""
    if not vocab._plain:           # a vocabulary with whitespace-bearing tokens that could fake a " ##" seam: plain composition
        return post_process(vocab.decode(ids, skip_special_tokens=True))
    import numpy as np
    a = np.asarray(list(ids) if not hasattr(ids, "__len__") else ids, dtype=np.int64).ravel()
    a = np.where((a >= 0) & (a < len(vocab.tokens)), a, len(vocab.tokens))
    a = a[~vocab._special[a]]
    if a.size == 0:
        return ""
    if _jaconv is not None or vocab._mark[a].any():
        text = vocab._first[a[0]] + "".join(vocab._rest[a[1:]].tolist())
        text = text.replace("…", "...")
        if "・" in text or ".." in text:
            text = re.sub("[・.]{2,}", lambda m: (m.end() - m.start()) * ".", text)
        return h2z(text, ascii=True, digit=True)
    text = vocab._first_pre[a[0]] + "".join(vocab._rest_pre[a[1:]].tolist())
    if "・" in text or ".." in text:
        text = re.sub("[・.]{2,}", lambda m: (m.end() - m.start()) * ".", text)
    if "." in text:
        text = text.replace(".", _H2Z_A["."])
    if "･" in text:
        text = text.replace("･", _H2Z_K["･"])
    return text


def find_vocab(model_dir: Optional[str]) -> Optional[str]:
    if model_dir:
        p = os.path.join(model_dir, "vocab.txt")
        if os.path.exists(p):
            return p
    return None
