#!/usr/bin/env python3
"""Instruction histogram of a kernel's longest loop from `llvm-objdump -d` of the gfx950 code object:
  /opt/rocm/lib/llvm/bin/llvm-objdump --offloading linrad_amd/liblinrad_hip.so   # extracts the device bundles
  /opt/rocm/lib/llvm/bin/llvm-objdump -d liblinrad_hip.so.0.hipv4-amdgcn-amd-amdhsa--gfx950 > k.s
  python3 scripts/isa_hist.py k.s _ZN3lrh7k_fft1vILi14ELb0ELb0ELi0ELb0EEEvNS_9Fft1wArgsE
(profiles/r06_fft1v_isa.txt)"""
import re, sys, collections
S = sys.argv[1]; name = sys.argv[2]
lines = open(S).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.endswith("<%s>:" % name))
end = next(i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i]))
body = lines[start + 1:end]
ins = []
for l in body:
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
    if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
# find backward branches
loops = []
for k, (addr, op, args) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"<[^>]*\+0x([0-9a-f]+)>|(\d+)\s*$", args)
        # objdump prints target as e.g. "s_cbranch_scc1 65432" or with symbol; compute from simm16
        mm = re.search(r"(-?\d+)\s*$", args.split("//")[0].strip())
        if mm:
            off = int(mm.group(1))
            if off >= 32768: off -= 65536
            tgt = addr + 4 + 4 * off
            if tgt < addr: loops.append((addr - tgt, tgt, addr))
loops.sort(reverse=True)
print("instructions:", len(ins), "loops (bytes, from, to):", [(a, hex(b), hex(c)) for a, b, c in loops[:5]])
if len(sys.argv) > 3 and sys.argv[3] == "all": lo, hi = 0, 1 << 60
else: _, lo, hi = loops[0]
sel = [(a, o, g) for a, o, g in ins if lo <= a <= hi]
def cls(op):
    if op.startswith("v_pk_"): return "VALU pk"
    if op.startswith("v_cvt"): return "VALU cvt"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")): return "VALU mov"
    if op.startswith(("v_add_u", "v_sub_u", "v_lshl", "v_lshr", "v_and", "v_or", "v_xor", "v_bfe", "v_mad_u", "v_mul_u", "v_mul_lo", "v_add_co", "v_addc", "v_ashr", "v_add3", "v_lshl_add", "v_mad_i", "v_bfi", "v_perm", "v_alignb", "v_sub_co", "v_subrev", "v_add_lshl", "v_lshl_or", "v_and_or", "v_or3", "v_xad", "v_min_i", "v_min_u", "v_max_i")): return "VALU int"
    if op.startswith(("v_cndmask", "v_cmp", "v_readlane", "v_readfirstlane", "v_writelane")): return "VALU sel/cmp"
    if op.startswith("v_"): return "VALU f32 scalar-lane"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_"): return "SALU"
    return "other"
c = collections.Counter(cls(o) for _, o, _ in sel)
tot = sum(c.values())
print("loop body: %d instructions" % tot)
for k, v in c.most_common(): print("  %-24s %6d  %5.1f%%" % (k, v, 100.0 * v / tot))
d = collections.Counter(o for _, o, _ in sel)
print("top mnemonics:")
for k, v in d.most_common(45): print("  %-28s %5d" % (k, v))
