"""TEST INFRASTRUCTURE ONLY -- tests/golden/cer_*.txt: stdout of the REAL reference scorer
(/root/reference/Multitask/utils/wenet_compute_cer.py, run as a subprocess exactly as Multitask/scripts/decode_sensevoice.sh:97
does) on the small gt/pred pairs written by tests/cer_fixtures.py.  Run in the build container only."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cer_fixtures as cf  # noqa: E402

SCRIPT = os.path.join(os.environ.get("TASU_REFERENCE_ROOT", "/root/reference"), "Multitask", "utils", "wenet_compute_cer.py")

if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        for name, flags in cf.CASES.items():
            gt, pred = cf.write_pair(d, name)
            extra = cf.write_side_files(d, name)
            out = subprocess.run([sys.executable, SCRIPT] + [f.format(**extra) for f in flags] + [gt, pred], check=True,
                                 capture_output=True, text=True, encoding="utf-8").stdout
            with open(os.path.join(ROOT, "tests", "golden", f"cer_{name}.txt"), "w", encoding="utf-8") as f:
                f.write(out)
            print(name, len(out.splitlines()), "lines;", [l for l in out.splitlines() if l.startswith("Overall")][0])
