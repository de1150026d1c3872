"""gt / pred decode-log pairs shared by oracle/make_golden_cer.py (runs the reference scorer) and tests/test_cer_cpu.py."""
import os

GT = ["utt01\tthe quick brown fox jumps over the lazy dog",
      "utt02\thello world",
      "utt03\t今天天气很好，我们去公园。",
      "utt04\tmixed 中文 and english words",
      "utt05\t<noise> it's nine o'clock <unk>",
      "utt06\tonly in reference",
      "utt07\ta b c d e f g",
      "utt08\tsame same",
      "utt09\t",
      "utt10\tcafé déjà vu",
      "utt11\tAAA bbb AAA bbb AAA"]
PRED = ["utt01\tthe quick brown fox jumped over lazy dog dog",
        "utt02\tHELLO WORLD!",
        "utt03\t今天天汽很好我去公园吧",
        "utt04\tmixed 中 and englsh words words",
        "utt05\tit's nine a clock",
        "utt07\tb c x e g h",
        "utt08\tsame same",
        "utt09\tspurious words",
        "utt10\tcafe deja vu",
        "utt11\tbbb AAA bbb AAA",
        "utt12\tonly in hypothesis"]
CASES = {   # name -> flags before the two file names
    "char_recipe": ["--char=1", "-v=1"],                     # the decode recipe's exact flags
    "word": [],
    "word_cs_quiet": ["--cs=1", "--v=0"],
    "char_v2_maxw": ["--char=1", "--v=2", "--maxw=4", "--padding-symbol=underline"],
    "word_ig_split_rt0": ["--ig={ig}", "--splitfile={split}", "--rt=0"],
}


def write_pair(d, name):
    gt, pred = os.path.join(d, f"{name}_gt"), os.path.join(d, f"{name}_pred")
    open(gt, "w", encoding="utf-8").write("\n".join(GT) + "\n")
    open(pred, "w", encoding="utf-8").write("\n".join(PRED) + "\n")
    return gt, pred


def write_side_files(d, name):
    ig, split = os.path.join(d, "ignore.txt"), os.path.join(d, "split.txt")
    open(ig, "w", encoding="utf-8").write("the\n<noise>\n")
    open(split, "w", encoding="utf-8").write("o'clock o clock\nenglsh eng lsh\n")
    return dict(ig=ig, split=split)
