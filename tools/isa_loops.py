"""Instruction mix of the loops of a gfx950 code object (llvm-objdump -d text on stdin or a file): for every backward branch
the [target, branch] range with its VALU / SALU / VMEM / LDS / other counts, largest first. Used to optimise the run-time
specialised kernels offline (the hiprtc compile needs no device)."""
import collections
import re
import sys

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
ins = []   # (addr, mnemonic, operands)
for line in text.splitlines():
    m = re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
    if m and "<" in m.group(2):
        m2 = re.match(r"\s+(\S+)\s+(\d+)", line)
        if m2:
            ins.append((int(m.group(3), 16), m.group(1), m2.group(2)))
            continue
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
loops = []
for i, (a, mn, ops) in enumerate(ins):
    if mn.startswith("s_cbranch") or mn == "s_branch":
        # the branch target: objdump prints it as a label or as an offset; compute from the simm16 when it is a number
        m = re.match(r"(\d+)", ops.strip())
        tgt = None
        if m:
            simm = int(m.group(1))
            simm = simm - 65536 if simm >= 32768 else simm
            tgt = a + 4 + simm * 4
        if tgt is not None and tgt in addr_index and tgt <= a:
            loops.append((addr_index[tgt], i))


def cls(mn):
    if mn.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    if mn.startswith("ds_"):
        return "LDS"
    if mn.startswith("v_"):
        return "VALU"
    if mn.startswith("s_waitcnt"):
        return "WAIT"
    if mn.startswith("s_load") or mn.startswith("s_buffer"):
        return "SMEM"
    if mn.startswith("s_"):
        return "SALU"
    return "OTHER"


loops.sort(key=lambda r: r[0] - r[1])
for lo, hi in loops[:6]:
    c = collections.Counter(cls(mn) for _, mn, _ in ins[lo:hi + 1])
    top = collections.Counter(mn for _, mn, _ in ins[lo:hi + 1]).most_common(14)
    print(f"loop {ins[lo][0]:#x}..{ins[hi][0]:#x}: {hi - lo + 1} instructions", dict(c))
    print("   ", ", ".join(f"{m}x{n}" for m, n in top))
