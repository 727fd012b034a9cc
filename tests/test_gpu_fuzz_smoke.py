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


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,workload", [(2, "soup100k"), (3, "raster4k")])
def test_bench_control_flow_of_several_ranks_on_one_gpu(ranks, workload):
    """bench.py --gpus N as the driver launches it for N > 1, rehearsed with every rank on device 0 (MIRT_BENCH_REHEARSAL=1: gloo group,
    host-staged gathers): the bands of the ranks (weighted where the frame is binned), the batches, the reductions -- and the frame
    rank 0 assembled compared word for word with a single-GPU render of the same view (bench.py exits non-zero when they differ).
    Not a measurement; at most 3 rank processes touch the card (the box allows 6)."""
    import json
    env = dict(os.environ, MIRT_BENCH_REHEARSAL="1", MIRT_BENCH_TARGET_S="0.1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--workload", workload, "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:] + r.stderr[-3000:])
    assert "identical to the single-GPU frame: True" in r.stderr, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == ranks and line["config"]["workload"] == workload and line["value"] > 0
