"""Instruction-class breakdown of a kernel's hot loop from its disassembly (developer tool).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only [-mllvm -amdgpu-mfma-vgpr-form] unit.hip -o unit.s
    python tools/isa_breakdown.py unit.s <mangled-kernel-substring> [--loop N]

Finds the kernel's function, its loops (a backward branch to a label), and prints for the N-th largest loop body
(default: the largest = the step loop of the persistent kernels) how many instructions of each class it holds:
matrix (v_mfma), Threefry-style integer VALU (add / xor / alignbit / shifts), transcendental (log, sqrt, sin, cos,
rcp), conversion, floating-point VALU (scalar and packed), moves / selects / compares, lane exchange (DPP, permlane,
readlane), LDS, vector memory, scalar ALU and control, waits and barriers.  A wave64 VALU instruction holds the SIMD
for 4 cycles (transcendentals 16, a 4x4x1 MFMA 8, 16x16x4 32, 32x32x2 64): the issue time of one trip follows."""
import re
import sys

CLASSES = [
    ("matrix (v_mfma)", r"^v_mfma"),
    ("transcendental", r"^v_(log|exp|sqrt|rsq|rcp|sin|cos)_"),
    ("conversion", r"^v_cvt_"),
    ("integer VALU (generator)", r"^v_(add_u32|add3_u32|sub_u32|xor|alignbit|lshl|lshr|ashr|and_b32|or_b32|or3|xad|add_co|addc_co|mul_lo|mul_hi|mad_u|lshl_add|lshl_or|and_or|bfe|bfi|xor3)"),
    ("packed fp32 VALU", r"^v_pk_"),
    ("fp32 VALU", r"^v_(fma|fmac|mul_f32|add_f32|sub_f32|subrev_f32|mac_f32|mad_f32|max_f32|min_f32|med3_f32|fract|floor|ldexp|frexp|max3|min3)"),
    ("lane exchange", r"^v_(permlane|readlane|readfirstlane|writelane|mov_b32_dpp|.*_dpp)|dpp|^ds_(bpermute|permute|swizzle)"),
    ("move / select / compare", r"^v_(mov|cndmask|cmp|cmpx|accvgpr|swap|nop)"),
    ("LDS", r"^ds_"),
    ("vector memory", r"^(global_|buffer_|flat_|scratch_)"),
    ("scalar memory", r"^s_(load|buffer_load|store|memtime|memrealtime)"),
    ("wait / barrier", r"^s_(waitcnt|barrier|sleep|nop|setprio)"),
    ("branch", r"^s_(cbranch|branch|endpgm|setpc|swappc)"),
    ("scalar ALU", r"^s_"),
]
CYCLES = {"transcendental": 16, "matrix (v_mfma)": None}


def mfma_cycles(op):
    if "4x4x" in op:
        return 8
    if "16x16x" in op:
        return 32
    return 64


def main():
    path, key = sys.argv[1], sys.argv[2]
    nth = int(sys.argv[sys.argv.index("--loop") + 1]) if "--loop" in sys.argv else 0
    lines = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", ln))
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, ln in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", ln))}
    loops = []
    for i, ln in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
    loops.sort(reverse=True)
    size, lo, hi = loops[nth]
    counts, cycles, unknown = {}, 0, {}
    for ln in body[lo:hi + 1]:
        ln = ln.split(";")[0].strip()
        if not ln or ln.startswith(".") or ln.endswith(":"):
            continue
        op = ln.split()[0]
        for name, pat in CLASSES:
            if re.search(pat, op if name != "lane exchange" else ln):
                counts[name] = counts.get(name, 0) + 1
                if name == "matrix (v_mfma)":
                    cycles += mfma_cycles(op)
                elif op.startswith("v_"):
                    cycles += CYCLES.get(name) or 4
                break
        else:
            unknown[op] = unknown.get(op, 0) + 1
    total = sum(counts.values()) + sum(unknown.values())
    print(f"{key}: loop of {total} instructions (lines {start + lo + 1}..{start + hi + 1} of {path}; "
          f"{len(loops)} loops in the function)")
    for name, _ in CLASSES:
        if counts.get(name):
            print(f"  {name:28s} {counts[name]:5d}")
    for op, n in sorted(unknown.items()):
        print(f"  (unclassified) {op:24s} {n:5d}")
    print(f"  vector-issue cycles of one trip (VALU 4, transcendental 16, MFMA 8 / 32 / 64): {cycles}")


if __name__ == "__main__":
    main()
