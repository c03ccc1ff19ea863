"""(CPU) Per-kernel opcode counts of a hipcc assembly dump: python tools/asm_kernel_stats.py file.s [name-substring]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n(_Z\S+):[^\n]*\n", txt)
KEYS = ["v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_mov_b32_e32",
        "s_nop", "ds_read_b128", "v_cvt_pk_f16_f32", "s_waitcnt", "scratch_load_dword", "scratch_store_dword", "s_barrier",
        "global_load_lds_dwordx4", "global_store_dwordx4", "global_load_dwordx4"]
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
    if flt not in name:
        continue
    ops = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line[0] in ".;/" or line.endswith(":"):
            continue
        ops[line.split()[0]] += 1
    print(name[:90], sum(ops.values()), {k.replace("v_mfma_f32_", "m").replace("_b32", ""): ops[k] for k in KEYS if ops[k]})
