"""Registers, LDS and scratch of every kernel in the built library (developer tool): reads the code-object metadata of
ccvm_amd/libccvm_hip.so (no GPU needed).
   python tools/kernel_resources.py [substring ...]      e.g.  persist_kernel  'step_kernel<0'"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib=os.path.join(ROOT, "ccvm_amd", "libccvm_hip.so")):
    """[{name, vgpr, agpr, sgpr, lds, scratch, spill}] of every kernel of the library's gfx950 code objects (one
    embedded ELF per translation unit inside the .hip_fatbin section)."""
    notes = ""
    data = open(lib, "rb").read()
    with tempfile.TemporaryDirectory() as tmp:
        pos, n = data.find(b"\x7fELF", 4), 0  # (offset 0 is the host library itself)
        while pos >= 0:
            path = os.path.join(tmp, f"elf{n}.o")
            with open(path, "wb") as fh:
                fh.write(data[pos:])
            r = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], capture_output=True, text=True)
            if ".vgpr_count" in r.stdout:
                notes += r.stdout
            pos, n = data.find(b"\x7fELF", pos + 4), n + 1
    found = {}
    for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        block = ".agpr_count:" + block
        get = lambda key: re.search(rf"\.{key}:\s+(\S+)", block)
        name = get("name")
        if name and name.group(1) not in found:
            found[name.group(1)] = {"vgpr": int(get("vgpr_count").group(1)), "agpr": int(get("agpr_count").group(1)),
                                    "sgpr": int(get("sgpr_count").group(1)),
                                    "lds": int(get("group_segment_fixed_size").group(1)),
                                    "scratch": int(get("private_segment_fixed_size").group(1)),
                                    "spill": int(get("vgpr_spill_count").group(1))}
    names = list(found)
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    except OSError:
        dem = names
    return [dict(found[m], name=(d or m)) for m, d in zip(names, dem)]


if __name__ == "__main__":
    want = sys.argv[1:]
    for k in sorted(kernels(), key=lambda k: k["name"]):
        if want and not any(w in k["name"] for w in want):
            continue
        print(f"{k['vgpr']:4d} vgpr {k['agpr']:4d} agpr {k['lds']:7d} B lds {k['scratch']:4d} B scratch {k['spill']:3d} spilled  {k['name']}")
