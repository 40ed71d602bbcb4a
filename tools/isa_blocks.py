"""Per-basic-block instruction counts of one kernel in an llvm .s dump, with the VALU issue-cycle class of every
instruction (2-cycle / 4-cycle on gfx950, from tools/valu_class_probe.hip).  usage: isa_blocks.py file.s kernel-substring"""
import re, sys
FAST = {"v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mul_legacy_f32", "v_not_b32"}
def cls(op, args):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith("_sdwa") or op.endswith("_dpp"): return 4
    if base in FAST:
        srcs = args.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[|vcc|exec|ttmp)", s) for s in srcs): return 4      # SGPR operand
        return 2
    return 4
def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(l.split(":")[0] + ":") or (l.startswith("_Z") and key in l and ":" in l))
    blocks, cur = [], None
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"): break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m or cur is None:
            cur = {"name": m.group(1) if m else "entry", "valu2": 0, "valu4": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "other": 0, "loop": ""}
            blocks.append(cur)
            if m and "Loop Header" in l: cur["loop"] = l.split(";")[-1].strip()
            if m: continue
        t = l.strip()
        if "Loop Header" in t or "in Loop" in t: cur["loop"] = cur["loop"] or t.lstrip("; ")
        m = re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_|flat_)(\S*)\s*(.*)", t)
        if not m: continue
        op = m.group(1) + m.group(2)
        if op.startswith("v_"):
            cur["valu%d" % cls(op, m.group(3).split(";")[0])] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer"): cur["smem"] += 1
        elif op.startswith("s_"): cur["salu"] += 1
        elif op.startswith("ds_"): cur["lds"] += 1
        else: cur["vmem"] += 1
    tot = {k: 0 for k in ("valu2", "valu4", "salu", "lds", "vmem", "smem")}
    for b in blocks:
        if b["valu2"] + b["valu4"] + b["salu"] + b["lds"] + b["vmem"] == 0: continue
        for k in tot: tot[k] += b[k]
        print(f'{b["name"]:12s} v2 {b["valu2"]:4d} v4 {b["valu4"]:4d} s {b["salu"]:4d} lds {b["lds"]:3d} vmem {b["vmem"]:3d} smem {b["smem"]:2d}  {b["loop"][:70]}')
    print("static total", tot)
main()
