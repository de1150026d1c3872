"""Scoring of the decode logs: error rate between ``<decode_log>_gt`` and ``<decode_log>_pred`` the way
the reference's recipe does it (Multitask/scripts/decode_sensevoice.sh:97: ``python utils/wenet_compute_cer.py --char=1 -v=1
gt pred``; the script is WeNet's compute-wer tool vendored at Multitask/utils/wenet_compute_cer.py).

    python -m ps_slm_amd.compute_cer --char=1 -v=1 decode_log_gt decode_log_pred > decode_log_wer

Same switches (``--char= --v= --cs= --rt= --maxw= --padding-symbol= --ig= --splitfile=``; anything else before the two file
names is skipped, so the recipe's ``-v=1`` is a no-op there as here), same tokenisation (:15-45), normalisation (:64-84),
alignment tie-breaking (deletion, then insertion, then diagonal on equal cost, :134-160) and the same report up to and
including the ``Overall ->`` line (:428-505).  Not reproduced: the per-cluster breakdown printed after it (:507-552).
Host-side text processing; no GPU work.  Parity: tests/golden/cer_*.txt are the reference script's own outputs
(oracle/make_golden_cer.py)."""
import sys
import unicodedata

import numpy as np

PUNCT = set("!,?、。！，；？：「」︰『』《》")
BLANKS = set(" \t\r\n")
DEL, INS, COR, SUB, START = 0, 1, 2, 3, 4


def characterize(text):
    """Character-mode tokens: every letter-other (CJK...) character alone, anything else as a run of ASCII up to blank (or up
    to and including '>' for '<tag>'); punctuation of the list, spaces and unassigned code points dropped."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        cat = unicodedata.category(c)
        if c in PUNCT or c in BLANKS or cat in ("Zs", "Cn"):
            i += 1
        elif cat == "Lo":
            out.append(c)
            i += 1
        else:
            stop = ">" if c == "<" else " "
            j = i + 1
            while j < n and ord(text[j]) < 128 and text[j] not in BLANKS and text[j] != stop:
                j += 1
            if j < n and text[j] == ">":
                j += 1
            out.append(text[i:j])
            i = j
    return out


def strip_tags(tok):
    out, i, n = [], 0, len(tok)
    while i < n:
        if tok[i] == "<":
            while i < n and tok[i] != ">":
                i += 1
            i += 1
        else:
            out.append(tok[i])
            i += 1
    return "".join(out)


def normalize(tokens, ignore=(), case_sensitive=False, split=None, remove_tag=True):
    out = []
    for t in tokens:
        if not case_sensitive:
            t = t.upper()
        if t in ignore:
            continue
        if remove_tag:
            t = strip_tags(t)
        if not t:
            continue
        if split and t in split:
            out += split[t]
        else:
            out.append(t)
    return out


def align(lab, rec):
    """Levenshtein alignment with unit costs.  -> (lab row, rec row, counts dict); '' marks the gap side."""
    L, R = len(lab), len(rec)
    dist = np.zeros((L + 1, R + 1), dtype=np.int64)
    back = np.full((L + 1, R + 1), START, dtype=np.int8)
    dist[:, 0], dist[0, :] = np.arange(L + 1), np.arange(R + 1)
    back[1:, 0], back[0, 1:] = DEL, INS
    for i in range(1, L + 1):
        li, drow, prow, brow = lab[i - 1], dist[i], dist[i - 1], back[i]
        for j in range(1, R + 1):
            best, how = prow[j] + 1, DEL
            c = drow[j - 1] + 1
            if c < best:
                best, how = c, INS
            same = li == rec[j - 1]
            c = prow[j - 1] + (0 if same else 1)
            if c < best:
                best, how = c, (COR if same else SUB)
            drow[j], brow[j] = best, how
    a, b, cnt = [], [], dict(all=0, cor=0, sub=0, ins=0)
    cnt["del"] = 0
    ops = []
    i, j = L, R
    while back[i, j] != START:
        how = int(back[i, j])
        if how in (COR, SUB):
            a.append(lab[i - 1]), b.append(rec[j - 1])
            i, j = i - 1, j - 1
        elif how == DEL:
            a.append(lab[i - 1]), b.append("")
            i -= 1
        else:
            a.append(""), b.append(rec[j - 1])
            j -= 1
        ops.append(how)
    a.reverse(), b.reverse(), ops.reverse()
    for how in ops:
        key = ("del", "ins", "cor", "sub")[how]
        cnt[key] += 1
        if how != INS:
            cnt["all"] += 1
    return a, b, cnt, ops


def width(s):
    return sum(1 + (unicodedata.east_asian_width(c) in "AFW") for c in s)


def rate(c):
    return (c["ins"] + c["sub"] + c["del"]) * 100.0 / c["all"] if c["all"] else 0.0


def counts_line(c):
    return "N=%d C=%d S=%d D=%d I=%d" % (c["all"], c["cor"], c["sub"], c["del"], c["ins"])


def read_table(path, tochar):
    for line in open(path, "r", encoding="utf-8"):
        arr = characterize(line) if tochar else line.strip().split()
        if arr:
            yield arr[0], arr[1:]


def score(ref_file, hyp_file, tochar=False, verbose=1, case_sensitive=False, remove_tag=True, ignore=(), split=None,
          max_words=sys.maxsize, pad=" ", out=None):
    """Writes the report to ``out`` (default stdout) and returns (overall counts, {utt: counts})."""
    out = out or sys.stdout
    w = out.write
    if not case_sensitive:
        ignore = {x.upper() for x in ignore}
        if split:
            split = {k.upper(): [x.upper() for x in v] for k, v in split.items()}
    hyp = {k: normalize(v, ignore, case_sensitive, split, remove_tag) for k, v in read_table(hyp_file, tochar)}
    per_token, per_utt = {}, {}

    def tok(t):
        return per_token.setdefault(t, dict(all=0, cor=0, sub=0, ins=0, **{"del": 0}))

    for key, toks in read_table(ref_file, tochar):
        if key not in hyp:
            continue
        lab = normalize(toks, ignore, case_sensitive, split, remove_tag)
        a, b, cnt, ops = align(lab, hyp[key])
        for t in lab + hyp[key]:
            tok(t)
        for x, y, how in zip(a, b, ops):
            name = ("del", "ins", "cor", "sub")[how]
            t = y if how == INS else x
            tok(t)[name] += 1
            if how != INS:
                tok(t)["all"] += 1
        per_utt[key] = cnt
        if verbose:
            w("\nutt: %s\n" % key)
            w("WER: %4.2f %% %s\n" % (rate(cnt), counts_line(cnt)))
            gaps = [max(width(x), width(y)) for x, y in zip(a, b)]
            lo = 0
            while lo < len(a):
                hi = min(len(a), lo + max_words)
                for tag, row in (("lab", a), ("rec", b)):
                    head = "%s(%s):" % (tag, key.encode("utf-8")) if verbose > 1 else tag + ":"
                    w(head + " " + "".join(t + pad * (g - width(t)) + " " for t, g in zip(row[lo:hi], gaps[lo:hi])) + "\n")
                w("\n")
                lo = hi
    if verbose:
        w("=" * 75 + "\n\n")
    total = dict(all=0, cor=0, sub=0, ins=0, **{"del": 0})
    for c in per_token.values():
        for k in total:
            total[k] += c[k]
    w("Overall -> %4.2f %% %s\n" % (rate(total), counts_line(total)))
    if not verbose:
        w("\n")
    return total, per_utt


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print("usage: python -m ps_slm_amd.compute_cer [--cs={0,1}] [--ig=ignore_file] [--char={0,1}] [--v={0,1,2}] "
              "[--padding-symbol={space,underline}] [--maxw=N] test.ref test.hyp > test.wer")
        return 0
    kw = dict(tochar=False, verbose=1, case_sensitive=False, remove_tag=True, ignore=set(), split=None, max_words=sys.maxsize, pad=" ")
    flag = lambda v: v.lower() == "true" or v.lower() != "0"
    while len(argv) > 2:
        a = argv.pop(0)
        name, _, val = a.partition("=")
        if name == "--maxw":
            kw["max_words"] = int(val)
        elif name == "--rt":
            kw["remove_tag"] = flag(val)
        elif name == "--cs":
            kw["case_sensitive"] = flag(val)
        elif name == "--char":
            kw["tochar"] = flag(val)
        elif name == "--v":
            try:
                kw["verbose"] = int(val)
            except ValueError:
                kw["verbose"] = 1 if flag(val) else 0
        elif name == "--padding-symbol":
            kw["pad"] = {"space": " ", "underline": "_"}.get(val.lower(), kw["pad"])
        elif name == "--ig":
            kw["ignore"] |= {l.strip() for l in open(val, encoding="utf-8") if l.strip()}
        elif name == "--splitfile":
            kw["split"] = {p[0]: p[1:] for p in (l.split() for l in open(val, encoding="utf-8")) if len(p) >= 2}
        # anything else (e.g. the recipe's "-v=1", "--cluster=") is skipped, as in the reference's option loop
    score(argv[0], argv[1], **kw)
    return 0


if __name__ == "__main__":
    sys.exit(main())
