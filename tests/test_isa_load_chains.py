"""CPU (hipcc cross-compiles gfx950 here): the strided pools must keep their loads in flight TOGETHER.  Round 5 found that hipcc
turns `inside ? load : 0` into a branch with a full `s_waitcnt vmcnt(0)` behind the load -- three to nine exposed memory round
trips per thread in these kernels, 117 instead of 84 us for MaxPool3d_2a.  The fix (clamped address + AND mask) is
easy to undo by accident, and nothing functional notices: this test reads the ISA the way tools/isa_loads.sh does and counts,
per kernel, how many times a FULL wait is followed by further global loads (= dependent round trips beyond the first)."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

# kernel-name fragment -> allowed "full wait, then more loads" transitions
BUDGET = {
    "maxpool133_s2_w8_fwd_kernel": 0,
    "maxpool133_s2_w8_nn_fwd_kernel": 0,
    "maxpool133_s2_w8_bwd_kernel": 0,
    "maxpoolk33_s2_fwd_kernelILi3ELb1ELb1": 0,          # <3, true, true>: MaxPool3d_4a forward on bf16 rows
    "maxpoolk33_s2_bwd_kernelILi3ELb1ELb1": 1,          # the sign byte's uniform branch
    "maxpool333_s2_w12_bwd_kernel": 2,
    "maxpool333_rows_bwd_kernelILi12ELb1": 0,
    "maxpool333_rows_bwd_kernelILi6ELb1": 0,
}


def _chains(asm):
    out, name, loads_after_full, waiting_full, seen_load = {}, None, 0, False, False
    for line in asm.splitlines():
        m = re.match(r"^(_Z\S+):\s+; @", line)
        if m:
            name, loads_after_full, waiting_full, seen_load = m.group(1), 0, False, False
            continue
        if name is None:
            continue
        if "global_load" in line or "buffer_load" in line:
            if waiting_full:
                loads_after_full += 1
                waiting_full = False
            seen_load = True
        elif "s_waitcnt" in line and "vmcnt(0)" in line and seen_load:
            waiting_full = True
        elif "s_endpgm" in line:
            out[name] = loads_after_full
            name = None
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_strided_pool_kernels_issue_their_loads_together(tmp_path):
    asm = os.path.join(str(tmp_path), "pool3d.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(REPO, "opental_amd", "csrc"), "-S", "--cuda-device-only",
                    os.path.join(REPO, "opental_amd", "csrc", "pool3d.hip"), "-o", asm], check=True, stderr=subprocess.DEVNULL)
    chains = _chains(open(asm).read())
    assert chains, "no kernels found in the ISA"
    for frag, budget in BUDGET.items():
        hits = {k: v for k, v in chains.items() if frag in k}
        assert hits, f"kernel {frag} not found"
        for k, v in hits.items():
            assert v <= budget, f"{k}: {v} dependent load rounds behind a full wait (budget {budget}): a conditional load crept back in?"
