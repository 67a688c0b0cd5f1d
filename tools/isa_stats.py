"""Kernel-tuning aid: register / scratch use and an instruction histogram of one kernel of a built object.
usage: python tools/isa_stats.py <object.o> <kernel-substring> [--dump out.s]"""
import collections
import os
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    obj, key = sys.argv[1], sys.argv[2]
    tmp = "/tmp/nfhip_isa"
    os.makedirs(tmp, exist_ok=True)
    fb, co = os.path.join(tmp, "x.fatbin"), os.path.join(tmp, "x.co")
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", obj], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    syms = re.findall(r"^[0-9a-f]+ <(\S+)>:$", dis, flags=re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
    want = [m for m, d in zip(syms, dem) if key in d]
    for m in want:
        d = dem[syms.index(m)]
        blk = [b for b in notes.split("- .agpr_count:")[1:] if f".name:           {m}" in b or m in b]
        g = lambda k, b: (re.search(rf"\.{k}:\s*(\S+)", b) or [None, "?"])[1]
        if blk:
            b = blk[0]
            print(f"{d[:140]}\n  agpr {b.split()[0]} vgpr {g('vgpr_count', b)} sgpr {g('sgpr_count', b)} scratch {g('private_segment_fixed_size', b)}")
        body = dis.split(f"<{m}>:\n", 1)[1].split("\n\n", 1)[0].splitlines()
        hist = collections.Counter(l.split()[0] for l in body if l.strip())
        cls = collections.Counter()
        for k, v in hist.items():
            c = ("mfma" if "mfma" in k else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("buffer_", "global_", "flat_", "scratch_")) else
                 "trans" if k.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq", "v_sin", "v_cos")) else "valu" if k.startswith("v_") else
                 "wait" if k.startswith("s_waitcnt") else "salu")
            cls[c] += v
        print("  total", len(body), dict(cls))
        print("  top:", ", ".join(f"{k} {v}" for k, v in hist.most_common(24)))
        if "--dump" in sys.argv:
            with open(sys.argv[sys.argv.index("--dump") + 1], "w") as f:
                f.write("\n".join(body))


if __name__ == "__main__":
    main()
