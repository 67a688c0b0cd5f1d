"""Registers, scratch and LDS of every kernel in the built objects (normalizingflows.jl_amd/build/*.o; from the code objects' metadata notes).
usage: python tools/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def kernel_table(bdir=None):
    """[(demangled name, agpr, vgpr, sgpr, scratch bytes, static LDS bytes)] of every kernel in the objects under `bdir`
    (default: the in-tree build directory).  Needs only the LLVM binutils of the ROCm image: runs in the build container."""
    tmp = "/tmp/nfhip_co"
    os.makedirs(tmp, exist_ok=True)
    bdir = bdir or os.path.join(ROOT, "normalizingflows.jl_amd", "build")
    rows = []
    for obj in sorted(f for f in os.listdir(bdir) if f.endswith(".o")):
        fb, co = os.path.join(tmp, obj + ".fatbin"), os.path.join(tmp, obj + ".co")
        r = subprocess.run([LLVM + "llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", os.path.join(bdir, obj)], capture_output=True)
        if r.returncode:
            continue
        subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        out = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in out.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
            rows.append((g("name"), blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    return [(n,) + tuple(int(x) if x.isdigit() else x for x in r[1:]) for n, r in zip(names, rows)]


def main():
    keys = sys.argv[1:]
    print(f"{'agpr':>5} {'vgpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>7}  kernel")
    for name, ag, vg, sg, sc, lds in sorted(kernel_table()):
        if keys and not any(k in name for k in keys):
            continue
        print(f"{ag:>5} {vg:>5} {sg:>5} {sc:>8} {lds:>7}  {name[:150]}")


if __name__ == "__main__":
    main()
