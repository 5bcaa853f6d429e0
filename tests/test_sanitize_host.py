"""CPU suite: the host C++ of the drop-in under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer
(tools/sanitize_host.sh: the readers' token buffer, the digit former, the text of rows and dumps formed by threads, the
partition plan, the update math, the shared-memory transport with its ranks as threads, and `cnF2freq --parse-only`).
Zero reports is the bar; the log of the run that was committed is profiles/r05_sanitizers_host.log."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_is_clean_under_asan_ubsan_and_tsan(tmp_path):
    import __graft_entry__ as g
    g.build()                                    # the sanitized executable links libcnf2hip.so
    log = tmp_path / "sanitizers.log"
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh"), str(log)], capture_output=True, text=True, timeout=900)
    text = log.read_text() if log.exists() else ""
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert text.count("sanitize_host: ok") == 2 and "zero reports" in text, text[-3000:]
    assert "3 passed" in text, text[-1500:]      # tests/test_cli_readers.py on the sanitized executable
