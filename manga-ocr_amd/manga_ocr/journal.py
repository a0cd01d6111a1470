"""Results journal: crop key -> decoded text, so that a long crop-job run (BASELINE configs[3]: a queue of 10,000 crops) can
be resumed after the process died instead of decoded again.

The reference persists its project file atomically - payload to ``<path>.tmp``, then ``os.replace``
(``src/core/workers.py:121-144``) - but never its in-flight job queue ("Checkpoint / resume").  This is
the inference-side counterpart: an append-only JSON-lines file, one self-contained record per finished crop, flushed
(and optionally fsync'ed) after every chunk of results.  A record torn by a crash is the file's LAST line and fails to
parse: it is dropped on load and the crop decoded again.  ``compact()`` rewrites the file with one record per key the
reference's way (``.tmp`` + ``os.replace``).

Host-side convenience only: nothing in the engine or the hot path depends on it.
"""
from __future__ import annotations

import json
import os
import threading
from typing import Callable, Dict, Hashable, Iterable, List, Optional, Sequence


class ResultsJournal:
    """``with ResultsJournal(path) as j: texts = j.run(keys, items, reader.recognize_batch)``"""

    def __init__(self, path: str, fsync: bool = False):
        self.path = path
        self.fsync = fsync
        self._lock = threading.Lock()
        self._done: Dict[str, str] = {}
        self.torn_records = 0
        if os.path.exists(path):
            with open(path, "r", encoding="utf-8") as fh:
                for line in fh:
                    line = line.rstrip("\n")
                    if not line:
                        continue
                    try:
                        rec = json.loads(line)
                        self._done[self._k(rec["key"])] = rec["text"]
                    except (ValueError, KeyError, TypeError):
                        self.torn_records += 1          # a record cut short by a crash: its crop is decoded again
        # never append behind half a line: a crash may have cut a record anywhere - including exactly in front of its trailing
        # newline, which leaves a record that parses and a file that does not end in one
        ends_clean = True
        if os.path.exists(path) and os.path.getsize(path) > 0:
            with open(path, "rb") as fb:
                fb.seek(-1, os.SEEK_END)
                ends_clean = fb.read(1) == b"\n"
        self._fh = open(path, "a", encoding="utf-8")
        if not ends_clean:
            self._fh.write("\n")
            self._fh.flush()

    # ------------------------------------------------------------------ container protocol
    @staticmethod
    def _k(key: Hashable) -> str:
        """In-memory key: always the JSON form, so that the int 1 ("1") and the str "1" ("\"1\"") stay two crops; a record
        stores the key as the JSON VALUE itself (a str key reads back as that str, a tuple as a list: the same form again)."""
        return json.dumps(key, sort_keys=True, ensure_ascii=False)

    def __contains__(self, key: Hashable) -> bool:
        return self._k(key) in self._done

    def __len__(self) -> int:
        return len(self._done)

    def get(self, key: Hashable, default: Optional[str] = None) -> Optional[str]:
        return self._done.get(self._k(key), default)

    # ------------------------------------------------------------------ writing
    def record(self, keys: Iterable[Hashable], texts: Iterable[str]) -> None:
        """Append one chunk of results; visible to a later process once this call returns."""
        with self._lock:
            for key, text in zip(keys, texts):
                k = self._k(key)
                self._done[k] = text
                self._fh.write(json.dumps({"key": json.loads(k), "text": text}, ensure_ascii=False) + "\n")
            self._fh.flush()
            if self.fsync:
                os.fsync(self._fh.fileno())

    def run(self, keys: Sequence[Hashable], items: Sequence, recognize: Callable[[List], Sequence[str]], chunk: int = 2048) -> List[str]:
        """Texts of ``items`` in order: the journal's for the keys it holds, ``recognize(chunk_of_items)`` for the rest, every
        chunk recorded as soon as it is decoded (a chunk that raises loses nothing recorded before it)."""
        if len(keys) != len(items):
            raise ValueError("one key per item")
        todo = [i for i, k in enumerate(keys) if k not in self]
        for s in range(0, len(todo), max(1, chunk)):
            part = todo[s:s + max(1, chunk)]
            texts = list(recognize([items[i] for i in part]))
            if len(texts) != len(part):
                raise RuntimeError(f"recognize returned {len(texts)} texts for {len(part)} items")
            self.record([keys[i] for i in part], texts)
        return [self._done[self._k(k)] for k in keys]

    def compact(self) -> None:
        """One record per key, written beside the journal and moved over it (``.tmp`` + ``os.replace``)."""
        with self._lock:
            tmp = self.path + ".tmp"
            with open(tmp, "w", encoding="utf-8") as fh:
                for k, text in self._done.items():
                    fh.write(json.dumps({"key": json.loads(k), "text": text}, ensure_ascii=False) + "\n")
                fh.flush()
                os.fsync(fh.fileno())
            self._fh.close()
            os.replace(tmp, self.path)
            self._fh = open(self.path, "a", encoding="utf-8")

    def close(self) -> None:
        with self._lock:
            if not self._fh.closed:
                self._fh.close()

    def __enter__(self) -> "ResultsJournal":
        return self

    def __exit__(self, *exc) -> None:
        self.close()
