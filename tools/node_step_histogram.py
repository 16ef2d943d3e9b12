#!/usr/bin/env python3
"""tools/node_step_histogram.py — instruction histogram of ONE node step of k_bvh<AO> (the dominant kernel), from the
compiler's assembly, priced with the issue rates tools/micro/valu_rate measured on MI355X (gpurun_out/r3a_valu_rate*.jsonl,
copied to profiles/r03_valu_rate.jsonl).  Answers VERDICT r2 item 1's fall-back: where the floor of the node step is.

    python tools/node_step_histogram.py > profiles/r03_node_step_histogram.txt
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracer_amd", "csrc")


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rt_bvh.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                        "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-S", "--cuda-device-only",
                        os.path.join(SRC, "rt_bvh.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    # the timed AO variant: k_bvh<1, false, 6>
    m = re.search(r"^(_Z5k_bvhILi1ELb0ELi6EE\w+):.*?^\s*s_endpgm", text, re.S | re.M)
    if not m:
        sys.exit("kernel k_bvh<1,false,6> not found in the assembly")
    body = m.group(0).split("\n")
    blocks, cur = [], []
    for line in body:
        if re.match(r"^\.LBB\d+_\d+:", line):
            blocks.append(cur); cur = [line]
        else:
            cur.append(line)
    blocks.append(cur)

    def insts(b):
        r = []
        for l in b:
            s = l.strip()
            if not s or s.startswith((";", ".")) or s.endswith(":"):
                continue
            r.append(s.split()[0])
        return r
    # node-step blocks = those holding the 24 v_perm_b32 of the four box tests (the repeat loop is unrolled: several copies)
    steps = [b for b in blocks if sum(1 for i in insts(b) if i == "v_perm_b32") == 24]
    if not steps:
        sys.exit("no block with 24 v_perm_b32 found")
    b = steps[0]
    # the step continues through the stack stores and the pop loop up to the next node-step / vote block: take the run of
    # blocks from this one to the next one that starts a step or the vote (first block with a ballot s_bcnt1 after it)
    i0 = blocks.index(b)
    run = [b]
    for nb in blocks[i0 + 1:]:
        ii = insts(nb)
        if "v_perm_b32" in ii or "s_bcnt1_i32_b64" in ii:
            break
        run.append(nb)
    rates = {}
    rpath = os.path.join(ROOT, "profiles", "r03_valu_rate.jsonl")
    if os.path.exists(rpath):
        for l in open(rpath):
            l = l.strip()
            if l.startswith("{"):
                d = json.loads(l)
                if d["waves_per_simd"] == 8:
                    rates[d["op"].split()[0]] = d["cycles_per_inst_per_simd"]
    default = rates.get("v_cndmask_b32_e64", 2.45)
    print("k_bvh<AO> (k_bvh<1, false, 6>): one node step = the block with the four box tests (%d instructions) and the %d blocks of stack\n"
          "stores / pop loop behind it (executed per lane as needed).  Issue cycles: tools/micro/valu_rate at 8 waves per SIMD.\n" % (len(insts(b)), len(run) - 1))
    for title, bl in (("box tests + child sort (straight line, every voting lane)", [b]), ("stack stores + pop loop (the blocks behind it, summed)", run[1:])):
        h = collections.Counter()
        for x in bl:
            h.update(insts(x))
        valu = {k: v for k, v in h.items() if k.startswith("v_")}
        cyc = 0.0
        print("== " + title)
        for k, v in sorted(h.items(), key=lambda kv: (-kv[1], kv[0])):
            base = re.sub(r"_e(32|64)$", "", k)
            r = rates.get(k) or rates.get(base)
            if k.startswith("v_"):
                c = (r if r else default) * v
                cyc += c
                print("  %-28s %4d   ~%5.1f cycles%s" % (k, v, c, "" if r else "  (rate of v_cndmask_b32_e64 assumed)"))
            else:
                print("  %-28s %4d" % (k, v))
        print("  VALU %d, SALU %d, LDS %d, VMEM %d; VALU issue ~%.0f cycles per wave-step and SIMD\n" % (
            sum(valu.values()), sum(v for k, v in h.items() if k.startswith("s_")), sum(v for k, v in h.items() if k.startswith("ds_")),
            sum(v for k, v in h.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))), cyc))


if __name__ == "__main__":
    main()
