"""The environment switchboard (round 4): one struct parsed once, < 25 documented switches, no getenv on the call path.
CPU test: reads the sources, the header's list through the C ABI (no GPU call) and INTEGRATION.md."""
import glob
import os
import re

from ndrustfft_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ndrustfft_amd", "csrc")


def _doc_table():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 6. Environment switches"):]
    return re.findall(r"^\| `(NDFFT_[A-Z0-9_]+)` \|", sec, re.M)


def test_documented_switches_match_the_library_and_the_docs():
    lib = _lib.default()
    names = lib.documented_switches()
    assert len(names) == len(set(names)) and 0 < len(names) < 25, names
    assert sorted(names) == sorted(_doc_table()), (sorted(set(names) ^ set(_doc_table())))
    # every documented switch is parsed by the one parser, and nothing else is
    plan = open(os.path.join(CSRC, "plan.hip")).read()
    parser = plan[plan.index("const Switches *parse_switches()"):plan.index("const Switches &sw()")]
    parsed = set(re.findall(r'"(NDFFT_[A-Z0-9_]+)"', parser))
    assert parsed == set(names), parsed ^ set(names)


def test_no_getenv_outside_the_parser():
    offenders = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))):
        src = open(f).read()
        if f.endswith("plan.hip"):
            a, b = src.index("const Switches *parse_switches()"), src.index("const Switches &sw()")
            src = src[:a] + src[b:]
        if f.endswith("switches.h"):
            src = re.sub(r"#ifdef NDFFT_DEV_KNOBS.*?#else", "", src, flags=re.S)      # the developer build's read-at-every-use macros
        for m in re.finditer(r"\bgetenv\s*\(", src):
            offenders.append((os.path.basename(f), src.count("\n", 0, m.start()) + 1))
    assert not offenders, offenders


def test_reload_is_exported_and_harmless():
    lib = _lib.default()
    os.environ["NDFFT_WAVE"] = "0"
    try:
        lib.reload_switches()
    finally:
        del os.environ["NDFFT_WAVE"]
        lib.reload_switches()
