"""CPU: the results journal (manga_ocr/journal.py) - resume of a crop-job run after a crash (the reference
persists its project file with .tmp + os.replace, src/core/workers.py:121-144, never its job queue)."""
import json
import os

import pytest

from manga_ocr.journal import ResultsJournal


def _recognizer(log):
    def rec(items):
        log.append(list(items))
        return [f"text<{it}>" for it in items]
    return rec


def test_run_decodes_only_what_the_journal_does_not_hold(tmp_path):
    path = str(tmp_path / "run.jsonl")
    keys = [("page1.png", i) for i in range(10)]
    items = list(range(100, 110))
    calls = []
    with ResultsJournal(path) as j:
        first = j.run(keys[:6], items[:6], _recognizer(calls), chunk=4)
    assert calls == [[100, 101, 102, 103], [104, 105]] and first == [f"text<{i}>" for i in items[:6]]
    calls.clear()
    with ResultsJournal(path) as j:                      # a new process: six crops are already there
        assert len(j) == 6 and keys[2] in j and keys[7] not in j
        out = j.run(keys, items, _recognizer(calls), chunk=3)
    assert calls == [[106, 107, 108], [109]]
    assert out == [f"text<{i}>" for i in items]


def test_a_record_torn_by_a_crash_is_dropped_and_decoded_again(tmp_path):
    path = str(tmp_path / "run.jsonl")
    with ResultsJournal(path) as j:
        j.record(["a", "b", "c"], ["あ", "い", "う"])
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:-9])                     # the last record loses its tail (power cut mid-write)
    calls = []
    with ResultsJournal(path) as j:
        assert j.torn_records == 1 and len(j) == 2 and j.get("a") == "あ" and "c" not in j
        out = j.run(["a", "b", "c", "d"], [1, 2, 3, 4], _recognizer(calls))
    assert calls == [[3, 4]] and out == ["あ", "い", "text<3>", "text<4>"]
    with ResultsJournal(path) as j:                      # the re-appended records parse: nothing was glued to the torn line
        assert j.torn_records == 1 and len(j) == 4


def test_a_failing_chunk_keeps_what_was_recorded_before_it(tmp_path):
    path = str(tmp_path / "run.jsonl")

    def rec(items):
        if 13 in items:
            raise RuntimeError("device lost")
        return [str(i) for i in items]

    with ResultsJournal(path) as j:
        with pytest.raises(RuntimeError):
            j.run(list(range(10, 16)), list(range(10, 16)), rec, chunk=2)
    with ResultsJournal(path) as j:
        assert sorted(j._done) == [json.dumps(10), json.dumps(11)]


def test_compact_rewrites_one_record_per_key_atomically(tmp_path):
    path = str(tmp_path / "run.jsonl")
    with ResultsJournal(path, fsync=True) as j:
        j.record(["k"], ["old"])
        j.record(["k", "m"], ["new", "x"])
        j.compact()
        assert not os.path.exists(path + ".tmp")
        j.record(["z"], ["after"])
    lines = [json.loads(line) for line in open(path, encoding="utf-8") if line.strip()]
    assert [(r["key"], r["text"]) for r in lines] == [("k", "new"), ("m", "x"), ("z", "after")]


def test_wrong_result_count_is_an_error(tmp_path):
    with ResultsJournal(str(tmp_path / "j.jsonl")) as j:
        with pytest.raises(RuntimeError):
            j.run([1, 2], ["a", "b"], lambda items: ["only one"])
        with pytest.raises(ValueError):
            j.run([1], ["a", "b"], lambda items: items)


def test_a_crash_that_cut_only_the_trailing_newline_loses_nothing(tmp_path):
    """r03 advisor: a complete last record without its newline parses - the next append must still start a new line, or the
    next load drops BOTH records."""
    from manga_ocr.journal import ResultsJournal
    p = str(tmp_path / "j.jsonl")
    with ResultsJournal(p) as j:
        j.record(["a", "b"], ["A", "B"])
    raw = open(p, "rb").read()
    assert raw.endswith(b"\n")
    open(p, "wb").write(raw[:-1])                      # the crash
    with ResultsJournal(p) as j:
        assert j.torn_records == 0 and j.get("b") == "B"
        j.record(["c"], ["C"])
    with ResultsJournal(p) as j:
        assert j.torn_records == 0 and [j.get(k) for k in "abc"] == ["A", "B", "C"]


def test_int_and_str_keys_do_not_shadow_each_other(tmp_path):
    from manga_ocr.journal import ResultsJournal
    p = str(tmp_path / "j.jsonl")
    with ResultsJournal(p) as j:
        j.record([1, "1", (2, "x")], ["int", "str", "tuple"])
    with ResultsJournal(p) as j:
        assert j.get(1) == "int" and j.get("1") == "str" and j.get((2, "x")) == "tuple" and len(j) == 3
