"""Scans the built gfx950 code objects for the wide-store hazard: a VMEM store of more than 64 bits followed DIRECTLY by a
VALU / MFMA instruction that writes one of its data registers.  hipcc inserts the wait state itself except for buffer
stores whose soffset is an SGPR (the ISA manual exempts them); on MI355X that form raced in k_affine_chain (DESIGN.md
section 5), so the kernels keep soffset = 0 on wide stores and this scan (tests/test_abi_cpu.py) keeps it that way.
usage: python tools/check_store_hazard.py   -> prints violations, exit code 1 if any"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
STORE = re.compile(r"^\s*((?:buffer|global|flat|scratch)_store_(?:dwordx[34]|format_xyzw?|b96|b128))\s+(?:v\d+|v\[\d+:\d+\]|off),?\s*(v\[(\d+):(\d+)\])?")
DATA = re.compile(r"_store_\S+\s+(?:v\d+,\s*|off,\s*)?v\[(\d+):(\d+)\]|_store_\S+\s+v\[(\d+):(\d+)\]")
DEST = re.compile(r"^\s*v_\S+\s+(?:v(\d+)|v\[(\d+):(\d+)\])")


def data_regs(line):
    # buffer_store_dwordx4 v[a:b], voff, s[..], soff   |   global_store_dwordx4 vaddr, v[a:b], s[..] / off
    regs = re.findall(r"v\[(\d+):(\d+)\]", line.split("//")[0])
    wide = [(int(a), int(b)) for a, b in regs if int(b) - int(a) >= 2]
    return wide[0] if wide else None


def scan(disasm):
    bad, kernel, prev = [], "?", None
    for line in disasm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel, prev = m.group(1), None
            continue
        text = line.split("//")[0].strip()
        if not text:
            continue
        if prev is not None:
            d = DEST.match(text)
            if d:
                lo = int(d.group(1) if d.group(1) is not None else d.group(2))
                hi = int(d.group(1) if d.group(1) is not None else d.group(3))
                if lo <= prev[1][1] and hi >= prev[1][0]:
                    bad.append((kernel, prev[0], text))
            prev = None
        if re.match(r"^(buffer|global|flat|scratch)_store_(dwordx[34]|format_xyzw?)\b", text):
            regs = data_regs(text)
            if regs:
                prev = (text, regs)
    return bad


def main():
    tmp = "/tmp/nfhip_co"
    os.makedirs(tmp, exist_ok=True)
    bdir = os.path.join(ROOT, "normalizingflows.jl_amd", "build")
    bad = []
    nstores = 0
    for obj in sorted(f for f in os.listdir(bdir) if f.endswith(".o")):
        fb, co = os.path.join(tmp, obj + ".fatbin"), os.path.join(tmp, obj + ".co")
        if subprocess.run([LLVM + "llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", os.path.join(bdir, obj)], capture_output=True).returncode:
            continue
        subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        dis = subprocess.run([LLVM + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout
        nstores += len(re.findall(r"_store_dwordx[34]", dis))
        bad += scan(dis)
    names = subprocess.run(["c++filt"], input="\n".join(b[0] for b in bad), capture_output=True, text=True).stdout.split("\n")
    for n, b in zip(names, bad):
        print(f"{n[:110]}\n    {b[1]}\n    {b[2]}")
    print(f"{nstores} wide stores scanned, {len(bad)} followed directly by a write of their data registers")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
