"""profiles/rNN_size_sweep.md from gpurun_out/size_sweep_{auto,tile}.txt (developer tool; tools/size_sweep.sh)."""
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


ran = {}  # what ccvm_describe_launch said ran (tools/time_small.py prints it since round 5)
NAMES = {"R": "persistent row-owner", "S": "column-slab persistent", "C": "column-cluster persistent",
         "P": "persistent tile kernel", "T1": "per-step tile kernel, 32 x 128 tiles", "T2": "per-step tile kernel, 32 x 64 tiles",
         "T4": "per-step tile kernel, 32 x 32 tiles"}


def parse(path):
    out = {}
    for line in open(path):
        # (round 6: "RU=auto PW=auto" and the kernel's name behind the family tag)
        m = re.match(r"(\w+):(\d+):(\d+)(:adam)?\s+RU=\w+(?:\s+PW=\w+)?\s+([\d.]+) us/step[^\[]*(?:\[([^\]]+)\])?", line.rstrip())
        if m:
            key = (m.group(1) + (" + Adam" if m.group(4) else ""), int(m.group(2)), int(m.group(3)))
            out[key] = float(m.group(5))
            if m.group(6) and path.endswith("_auto.txt"):  # (the second pass forces the per-step kernel: not what runs by default)
                ran[key] = m.group(6)
    return out


auto, tile = parse("gpurun_out/size_sweep_auto.txt"), parse("gpurun_out/size_sweep_tile.txt")
flops = {"dl": 4, "mf": 2, "langevin": 2, "pl": 2, "mf + Adam": 2, "langevin + Adam": 2}


def path(n):
    if 64 < n <= 128:
        return "persistent row-owner (K split over two waves per SIMD)"
    if 128 < n <= 256:
        return "persistent row-owner (K split where the unsplit kernel leaves one wave per SIMD: N > ~200)"
    if n <= 256:
        return "persistent row-owner"
    if n <= 512:
        return "column-cluster persistent"
    if n <= 640:
        return "column-cluster persistent (3 row sets, k >= 512 in registers)"
    if n <= 768:
        return "column-cluster persistent (3 row sets, spread over the XCDs)"
    if 1024 < n <= 1536 or 2048 < n <= 2560:
        return "per-step tile kernel (32 x 64 tiles: fewer CU rounds)"
    return "per-step tile kernel"


print(f"# Round {tag[1:].lstrip('0')}: per-step time and fraction of the fp32 MFMA peak across problem sizes (B = 1000, 1x MI355X)\n")
print("`tools/size_sweep.sh` = `python3 tools/time_small.py kind:N:1000 ...` (4096-step chunks of the engine, best of 5, no "
      "profiler) with the default kernel policy and, for 256 < N <= 768, with `CCVM_AMD_KERNEL=nocluster` (last column). "
      "Algorithmic flops per step: DL 4 N^2 B, MF / Langevin / PL 2 N^2 B; peak 157.3 TFLOP/s.\n")
print("| solver | N | B | kernel path | us/step | TFLOP/s | frac of 157.3 | tile kernel at this size (us/step, frac) |")
print("|---|---|---|---|---|---|---|---|")
for (k, n, b), t in sorted(auto.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    tf = flops[k] * n * n * b / (t * 1e-6) / 1e12
    tl = tile.get((k, n, b))
    extra = "" if tl is None else f"{tl:.2f}, {flops[k] * n * n * b / (tl * 1e-6) / 1e12 / 157.3:.2f}"
    fam = ran.get((k, n, b))
    what = path(n) if fam is None else " + ".join(NAMES.get(f, NAMES.get(f[:1], f) + (f" ({f[1:]} slices)" if f[:1] == "P" and f[1:] else "")) for f in fam.split("+"))
    print(f"| {k} | {n} | {b} | {what} | {t:.2f} | {tf:.1f} | {tf / 157.3:.2f} | {extra} |")
