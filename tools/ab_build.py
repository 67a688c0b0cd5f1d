"""Build a VARIANT of libnfhip.so for same-box A/B measurements: `python tools/ab_build.py <name> [-DFLAG ...]` compiles every
translation unit with the extra flags into normalizingflows.jl_amd/build_ab/<name>/ and links normalizingflows.jl_amd/ab/<name>.so
(git-ignored like the shipped library, but it travels to the GPU box).  On the box an A/B script copies a variant over
normalizingflows.jl_amd/libnfhip.so between runs (the box's copy is scratch).  Only files whose text mentions one of the
flags' macro names are recompiled with them; the others reuse the shipped objects."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    macros = [re.sub(r"^-D", "", f).split("=")[0] for f in flags if f.startswith("-D")]
    objdir = os.path.join(ge.PKG_DIR, "build_ab", name)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.join(ge.PKG_DIR, "ab"), exist_ok=True)
    ge.build()  # the shipped objects must be current
    hdr_text = {h: open(os.path.join(ge.CSRC, h)).read() for h in os.listdir(ge.CSRC) if h.endswith(".h")}

    def uses(src):
        text = open(src).read()
        for h, t in hdr_text.items():
            if f'"{h}"' in text:
                text += t
        # one more level of includes (nf_mfma.h is pulled in through other headers)
        for h, t in hdr_text.items():
            if f'"{h}"' in text:
                text += t
        return any(m in text for m in macros)

    def one(s):
        src = os.path.join(ge.CSRC, s)
        if not uses(src):
            return os.path.join(ge.PKG_DIR, "build", s.replace(".hip", ".o"))
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        subprocess.run([ge.HIPCC, *ge.FLAGS, *flags, "-c", src, "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, ge.SOURCES))
    out = os.path.join(ge.PKG_DIR, "ab", name + ".so")
    subprocess.run([ge.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    print(out, "rebuilt:", [o for o in objs if "build_ab" in o])


if __name__ == "__main__":
    main()
