"""The plugin's host-side `.spz` parser (rcppml_amd/csrc/spz_parse.hpp -- the code that reads untrusted file bytes before the
device decoder starts) compiled for the CPU with ASan + UBSan and driven with the reference-written fixtures and a corpus of
truncated / corrupted copies (tools/sanitize/spz_parse_fuzz.cpp).  The full pass, including the oracle rebuilt under the
sanitizers, is tools/sanitize/run.sh (output committed as profiles/r03_sanitizer.txt)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spz_host_parser_under_asan_ubsan(tmp_path):
    exe = tmp_path / "spz_parse_fuzz"
    cc = subprocess.run(["g++", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g",
                         "-O1", os.path.join(ROOT, "tools", "sanitize", "spz_parse_fuzz.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert cc.returncode == 0, cc.stderr[-2000:]
    z = np.load(os.path.join(ROOT, "tests", "golden", "spz_vectors.npz"), allow_pickle=False)
    files = []
    for k in z.files:
        a = z[k]
        if a.dtype == np.uint8 and a.ndim == 1 and a.size >= 128 and bytes(a[:4]) == b"SPRZ":
            a.tofile(tmp_path / (k + ".spz"))
            files.append(str(tmp_path / (k + ".spz")))
    assert len(files) >= 8
    run = subprocess.run([str(exe), *files], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "no sanitizer report" in run.stdout and "refused" in run.stdout
