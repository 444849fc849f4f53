"""Measurement aid: registers, LDS and scratch of the kernels in a built library (the AMDGPU metadata notes of its code
object), e.g.  python3 tools/kernel_resources.py sbm-bp_amd/csrc/variants/libsbmbp_w3.so k_wsweep"""
import re
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def resources(lib, pattern=""):
    data = open(lib, "rb").read()
    out = []
    for m in list(re.finditer(b"\x7fELF", data))[1:]:  # the first ELF header is the host library itself
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(data[m.start():])
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = {k: re.search(r"\.%s:\s+(\S+)" % k, blk) for k in ("name", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size")}
            if not g["name"] or pattern not in g["name"].group(1):
                continue
            agpr = blk.split()[0]
            out.append((g["name"].group(1), int(g["vgpr_count"].group(1)), int(agpr), int(g["group_segment_fixed_size"].group(1)), int(g["private_segment_fixed_size"].group(1))))
    return out


if __name__ == "__main__":
    for name, vgpr, agpr, lds, scratch in resources(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-60s vgpr %3d (of them agpr %3d)  lds %6d  scratch %d" % (demangled[-60:], vgpr, agpr, lds, scratch))
