"""Kernel-tuning aid: the instruction mix of a kernel BETWEEN its barriers / branches, from a dump written by
tools/isa_stats.py --dump.  usage: python tools/isa_segments.py dump.s [--min N]"""
import collections
import sys


def cls(k):
    return ("mfma" if "mfma" in k else "lds" if k.startswith("ds_") else "scratch" if k.startswith("scratch_") else
            "vmem" if k.startswith(("buffer_", "global_")) else
            "trans" if k.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq")) else "valu" if k.startswith("v_") else
            "wait" if k.startswith("s_waitcnt") else "nop" if k.startswith("s_nop") else "salu")


def main():
    lines = [l.strip() for l in open(sys.argv[1]) if l.strip()]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 20
    cur, start = collections.Counter(), 0
    for i, l in enumerate(lines + ["s_barrier"]):
        op = l.split()[0]
        if op == "s_barrier" or op.startswith(("s_cbranch", "s_branch")):
            if sum(cur.values()) >= mn:
                print(f"{start:5d}-{i:5d} {dict(cur)}")
            if op == "s_barrier":
                print(f"{i:5d} s_barrier")
            cur, start = collections.Counter(), i + 1
        else:
            cur[cls(op)] += 1


if __name__ == "__main__":
    main()
