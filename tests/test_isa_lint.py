"""CPU (no GPU): properties of the compiled gfx950 code that the source alone does not guarantee.

ADVICE r5: the last-arriver finalize (voicemap_amd/csrc/bnpool.hip ``last_arriver``) is only correct if every wave has its
write-through (sc1) partial stores ACKNOWLEDGED before the workgroup's barrier and the ticket atomic -- an explicit
``s_waitcnt vmcnt(0)``; a workgroup-scope release fence alone compiled to store / s_barrier / atomic back to back.  hipcc
cross-compiles without a GPU, so the instruction order is checked here on the device assembly."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _device_asm(src, tmp_path):
    out = str(tmp_path / (os.path.basename(src) + ".s"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S",
           os.path.join(ROOT, "voicemap_amd", "csrc", src), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, cwd=str(tmp_path))
    with open(out) as f:
        return f.read().splitlines()


@pytest.mark.skipif(not os.path.exists(HIPCC) and shutil.which("hipcc") is None, reason="no hipcc")
def test_last_arriver_waits_for_its_partial_stores_before_the_ticket(tmp_path):
    lines = _device_asm("bnpool.hip", tmp_path)
    tickets = [i for i, l in enumerate(lines) if re.match(r"\s*global_atomic_add\S*\s", l) and "sc0" in l]   # the returning ticket fetch_add
    assert tickets, "no ticket atomic found: did last_arriver change?"
    checked = 0
    for t in tickets:
        # walk back to the barrier in front of the ticket, then on to the last sc1 store: a full vmcnt(0) wait must lie between
        j = t
        while j > 0 and "s_barrier" not in lines[j]:
            j -= 1
        assert j > 0, "ticket atomic without a barrier in front of it"
        k = j
        while k > 0 and not (re.match(r"\s*global_store", lines[k]) and "sc1" in lines[k]):
            k -= 1
            if re.match(r"^\S+:\s*;\s*@", lines[k]) or ".amdhsa_kernel" in lines[k]:   # start of the function: no store precedes
                k = 0
        if k == 0:
            continue
        between = [l.split(";")[0].strip() for l in lines[k + 1:j]]
        assert any(re.fullmatch(r"s_waitcnt vmcnt\(0\)( .*)?", b) or b == "s_waitcnt vmcnt(0)" for b in between), \
            "sc1 partial stores can still be in flight at the ticket (line %d)" % (t + 1)
        checked += 1
    assert checked >= 4    # the BatchNorm statistics / backward sums / column sums / tap sums instantiations
