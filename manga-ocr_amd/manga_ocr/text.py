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

_HALF_ASCII = "!\"#$%&'()*+,-./:;<=>?@[\\]^_`~ " + "abcdefghijklmnopqrstuvwxyz" + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "{|}"
_FULL_ASCII = "！＂＃＄％＆＇（）＊＋，－．／：；＜＝＞？＠［＼］＾＿｀～　" + "ａｂｃｄｅｆｇｈｉｊｋｌｍｎｏｐｑｒｓｔｕｖｗｘｙｚ" + \
              "ＡＢＣＤＥＦＧＨＩＪＫＬＭＮＯＰＱＲＳＴＵＶＷＸＹＺ" + "｛｜｝"
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


def h2z(text: str, kana: bool = True, ascii: bool = False, digit: bool = False) -> str:
    """Half-width -> full-width (jaconv.h2z semantics: voiced / semi-voiced marks combine with
    the preceding half-width katakana)."""
    out: List[str] = []
    i, n = 0, len(text)
    while i < n:
        ch = text[i]
        if kana:
            pair = text[i:i + 2]
            if pair in _H2Z_V:
                out.append(_H2Z_V[pair]); i += 2; continue
            if pair in _H2Z_S:
                out.append(_H2Z_S[pair]); i += 2; continue
            if ch in _H2Z_K:
                out.append(_H2Z_K[ch]); i += 1; continue
        if ascii and ch in _H2Z_A:
            out.append(_H2Z_A[ch])
        elif digit and ch in _H2Z_D:
            out.append(_H2Z_D[ch])
        else:
            out.append(ch)
        i += 1
    return "".join(out)


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
        toks = []
        for i in ids:
            i = int(i)
            if skip_special_tokens and i in self.special_ids:
                continue
            toks.append(self.tokens[i] if 0 <= i < len(self.tokens) else "[UNK]")
        return " ".join(toks).replace(" ##", "").strip()


def ids_to_text(vocab: Vocab, ids: Iterable[int]) -> str:
    return post_process(vocab.decode(ids, skip_special_tokens=True))


def find_vocab(model_dir: Optional[str]) -> Optional[str]:
    if model_dir:
        p = os.path.join(model_dir, "vocab.txt")
        if os.path.exists(p):
            return p
    return None
