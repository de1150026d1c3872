"""ps_slm_amd.compute_cer against the reference scorer's own stdout (tests/golden/cer_*.txt, oracle/make_golden_cer.py):
every line up to and including 'Overall ->' must be identical."""
import io
import os

import pytest

import cer_fixtures as cf
from ps_slm_amd import compute_cer

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def upto_overall(text):
    lines = text.split("\n")
    k = max(i for i, l in enumerate(lines) if l.startswith("Overall ->"))
    return lines[: k + 1]


@pytest.mark.parametrize("name", list(cf.CASES))
def test_report_matches_reference_scorer(tmp_path, name, capsys):
    gt, pred = cf.write_pair(str(tmp_path), name)
    extra = cf.write_side_files(str(tmp_path), name)
    compute_cer.main([f.format(**extra) for f in cf.CASES[name]] + [gt, pred])
    got = capsys.readouterr().out
    want = open(os.path.join(GOLDEN, f"cer_{name}.txt"), encoding="utf-8").read()
    assert upto_overall(got) == upto_overall(want)


def test_alignment_tie_breaking_and_counts():
    a, b, cnt, _ = compute_cer.align(list("abc"), list("abc"))
    assert cnt == dict(all=3, cor=3, sub=0, ins=0, **{"del": 0})
    _, _, cnt, _ = compute_cer.align([], ["x", "y"])
    assert cnt["ins"] == 2 and cnt["all"] == 0 and compute_cer.rate(cnt) == 0.0
    _, _, cnt, _ = compute_cer.align(["x", "y"], [])
    assert cnt["del"] == 2 and compute_cer.rate(cnt) == 100.0


def test_score_returns_per_utterance_counts(tmp_path):
    gt, pred = cf.write_pair(str(tmp_path), "api")
    total, per = compute_cer.score(gt, pred, tochar=True, out=io.StringIO())
    assert "utt06" not in per and "utt12" not in per and per["utt08"]["cor"] == 2
    assert total["all"] == sum(c["all"] for c in per.values())
