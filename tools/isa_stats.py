#!/usr/bin/env python
"""Static instruction statistics of the gfx950 kernels in one .hip file (no GPU needed):
    python tools/isa_stats.py [multiply_amd/csrc/mlp.hip] [-D...]
Per kernel: instructions, MFMAs, instructions per MFMA, branches, full LDS drains (s_waitcnt lgkmcnt(0)) and partial
waits, s_nop, register moves, VGPRs, code bytes.  The fused MLP kernels are bound by issue slots and by waves parked
on s_waitcnt, so these counts are what a change to mlp_core.hpp should be judged by before it goes to the GPU."""
import collections
import os
import re
import subprocess
import sys
import tempfile

src = next((a for a in sys.argv[1:] if not a.startswith("-")), "multiply_amd/csrc/mlp.hip")
flags = [a for a in sys.argv[1:] if a.startswith("-")]
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src] + flags,
               check=True, stderr=subprocess.DEVNULL)
text = open(out).read().split("\n")
starts = [(i, m.group(1)) for i, l in enumerate(text) if (m := re.match(r"^(_Z\w+):", l))]
meta = {}
cur = None
for l in text:
    m = re.match(r"\s+\.name:\s+(_Z\w+)$", l) or re.match(r"\s+- \.name:\s+(_Z\w+)$", l)
    if m:
        cur = m.group(1)
    m = re.match(r"\s+\.vgpr_count:\s+(\d+)", l)
    if m and cur:
        meta.setdefault(cur, {})["vgpr"] = int(m.group(1))
    m = re.match(r"\s+\.sgpr_spill_count:\s+(\d+)", l)
    if m and cur:
        meta.setdefault(cur, {})["sspill"] = int(m.group(1))
print(f"{'kernel':28s} {'instr':>6s} {'mfma':>5s} {'i/mfma':>6s} {'branch':>6s} {'lgkm0':>5s} {'lgkmN':>5s} {'vm0':>4s} {'s_nop':>5s} "
      f"{'mov':>5s} {'cndm':>5s} {'vgpr':>4s} {'sspl':>4s}")
for k, (i0, name) in enumerate(starts):
    i1 = starts[k + 1][0] if k + 1 < len(starts) else len(text)
    c = collections.Counter()
    lg0 = lgn = vm0 = 0
    for l in text[i0:i1]:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        if l.startswith(".") or l.startswith("s_endpgm"):
            continue
        op = l.split()[0]
        c[op] += 1
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                if int(m.group(1)) == 0:
                    lg0 += 1
                else:
                    lgn += 1
            if "vmcnt(0)" in l:
                vm0 += 1
        if op == "s_endpgm":
            break
    tot = sum(c.values())
    mf = sum(v for o, v in c.items() if "mfma" in o)
    br = sum(v for o, v in c.items() if o.startswith("s_cbranch") or o == "s_branch")
    mov = c["v_mov_b32_e32"] + c["v_mov_b64_e32"] + c["v_accvgpr_read_b32"] + c["v_accvgpr_write_b32"]
    short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:28]
    md = meta.get(name, {})
    print(f"{short:28s} {tot:6d} {mf:5d} {tot / max(mf, 1):6.1f} {br:6d} {lg0:5d} {lgn:5d} {vm0:4d} {c['s_nop']:5d} {mov:5d} "
          f"{c['v_cndmask_b32_e64'] + c['v_cndmask_b32_e32']:5d} {md.get('vgpr', -1):4d} {md.get('sspill', -1):4d}")
