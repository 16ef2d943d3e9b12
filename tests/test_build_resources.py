"""Register / scratch budget of the hot kernels, read from the compiler's own report (make resource-usage: hipcc
cross-compiles gfx950 here, no GPU needed).  The traversal and gate-walk kernels are latency-sensitive — their time follows
the waves that fit ("Occupancy") — and twice in round 3 an innocent-looking edit made the compiler keep the
kernels' by-value scene argument in scratch memory (560 bytes per lane: k_ao 34 -> 65 ms) without any test noticing."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_hot_kernels_keep_their_occupancy_and_stay_out_of_scratch():
    out = subprocess.run(["make", "-s", "resource-usage"], cwd=os.path.join(ROOT, "raytracer_amd", "csrc"),
                         capture_output=True, text=True, timeout=900).stdout
    kernels = {}
    name = None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            kernels[name][m.group(1).split(" ")[0]] = int(m.group(2))
    assert kernels, out[-2000:]

    def one(prefix):
        hits = [v for k, v in kernels.items() if k.startswith(prefix)]
        assert len(hits) == 1, (prefix, [k for k in kernels if k.startswith(prefix[:12])])
        return hits[0]

    # (mangled-name prefix, waves per SIMD at least, scratch bytes per lane at most) — the timed (STATS = false) variants
    budget = [
        ("_Z5k_bvhILi1ELb0ELi6E", 6, 0),    # k_bvh<AO>, the dominant kernel
        ("_Z5k_bvhILi0ELb0ELi6E", 6, 0),    # k_bvh<closest>
        ("_Z5k_bvhILi2ELb0ELi6E", 6, 0),    # k_bvh<shadow>
        ("_Z4k_aoILb0ELb0ELi6E", 6, 0),     # k_ao, per-lane walk (fewer than 4 gate nodes behind the room planes)
        ("_Z4k_aoILb0ELb1ELi7E", 7, 0),     # k_ao<WAVE>, the walk in step (config 3)
        ("_Z4k_aoILb0ELb1ELi8E", 8, 16),    # ... over 12 gate nodes or more (configs 1, 2, 4): 3 spilled dwords (the AO frame held across the strata) measure the same as none
        ("_Z9k_ao_walkILb0E", 8, 0),        # k_ao_walk (config 5)
        ("_Z7k_traceILb0E", 6, 0),
        ("_Z14k_shadow_deferILb0E", 6, 0),
    ]
    for prefix, waves, scratch in budget:
        k = one(prefix)
        assert k["Occupancy"] >= waves, (prefix, k)
        assert k["ScratchSize"] <= scratch, (prefix, k)
    # six blocks of k_bvh per CU: 160 KB of LDS
    assert one("_Z5k_bvhILi1ELb0ELi6E")["LDS"] * 6 <= 160 * 1024
    # ... and eight of the gate-walking kernels
    assert one("_Z4k_aoILb0ELb1ELi8E")["LDS"] * 8 <= 160 * 1024
    assert one("_Z9k_ao_walkILb0E")["LDS"] * 8 <= 160 * 1024
