"""The scripts under tools/ run on the GPU box only, at the end of a round, when a syntax error costs a GPU call: parse them all here."""
import glob
import os
import py_compile
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shell_scripts_parse():
    scripts = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert scripts
    for s in scripts:
        r = subprocess.run(["bash", "-n", s], capture_output=True, text=True)
        assert r.returncode == 0, (s, r.stderr)


def test_python_tools_compile(tmp_path):
    tools = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for i, t in enumerate(tools):
        py_compile.compile(t, cfile=str(tmp_path / ("%d.pyc" % i)), doraise=True)


def test_collection_scripts_name_existing_tools():
    """tools/collect_profiles.sh, final_collect.sh and fuzz_campaign.sh call other files of tools/ by name."""
    import re
    for name in ("collect_profiles.sh", "final_collect.sh", "fuzz_campaign.sh", "rehearse_ranks.sh"):
        text = open(os.path.join(ROOT, "tools", name)).read()
        for ref in set(re.findall(r"tools/([A-Za-z0-9_]+\.(?:py|sh))", text)):
            assert os.path.exists(os.path.join(ROOT, "tools", ref)), (name, ref)


def test_reference_build_products_stay_out_of_history_but_travel():
    """oracle/_ref/ (the reference's functions compiled in the build container) is git-ignored, so no reference product enters
    the history, and NOT gpurun-ignored, so the built libraries reach the GPU box like the other built .so files."""
    ignored = [l.strip() for l in open(os.path.join(ROOT, ".gitignore")) if l.strip() and not l.startswith("#")]
    assert "oracle/_ref/" in ignored
    p = os.path.join(ROOT, ".gpurunignore")
    if os.path.exists(p):
        assert not any("oracle/_ref" in l for l in open(p) if not l.startswith("#"))
    r = subprocess.run(["git", "-C", ROOT, "ls-files", "oracle/_ref"], capture_output=True, text=True)
    if r.returncode == 0:           # an exported tree has no .git: nothing to check there
        assert r.stdout.strip() == ""
