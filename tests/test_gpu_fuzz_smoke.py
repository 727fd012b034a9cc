"""The call-sequence fuzzers of tools/ inside the driver's test run: a few dozen seeded sequences each (seconds), so that the state
machine around the kernels -- light-cube cache and per-frame light pass, guessed list sizes, kept binning passes, cull flags per
stream, scratch growth, one to four frames in flight -- is walked on every round's final code, not only when the builder
remembers to.  The long runs (hundreds of sequences) stay in tools/collect_profiles.sh fuzz; their summaries are under profiles/."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args", [("fuzz_sequence.py", ["1000", "20"]), ("fuzz_raster_sequence.py", ["1000", "20"]),
                                       ("fuzz_binned.py", ["1000", "20", "6"]), ("fuzz_small.py", ["1000", "150"])])
def test_fuzzer_finds_no_mismatch(tool, args):
    """One child process per fuzzer (each initialises the library itself), run one after the other."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, (r.stdout[-3000:] + r.stderr[-3000:])
