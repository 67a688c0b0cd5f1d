"""Diff of two parity tables (gpurun_out/parity_measured.json of two `pytest -m gpu` runs on one box, e.g. the default
arithmetic and NF_FWD_FP32=1 NF_BWD_FP32=1 NF_WIDE_FP32=1): per check the ratio a / b, sorted; the geometric mean over all
checks says whether one arithmetic is worse across the board (1.0 = same), the two tails are the individual draws.
usage: python tools/parity_diff.py a.json b.json [--top 20] [--label-a default --label-b fp32_mfma]"""
import argparse
import json

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a"), ap.add_argument("b")
    ap.add_argument("--top", type=int, default=20)
    ap.add_argument("--label-a", default="a"), ap.add_argument("--label-b", default="b")
    ap.add_argument("--keys", nargs="*", default=[])
    args = ap.parse_args()
    a, b = json.load(open(args.a)), json.load(open(args.b))
    rows = [(a[k] / b[k], a[k], b[k], k) for k in a if k in b and "floor" not in k and "oracle" not in k and a[k] > 0 and b[k] > 0]
    rows.sort(reverse=True)
    r = np.array([x[0] for x in rows])
    print(f"{len(rows)} checks in both tables ({len(a)} / {len(b)} entries); ratio = {args.label_a} / {args.label_b}")
    print(f"geometric mean of the ratio {np.exp(np.log(r).mean()):.3f}; median {np.median(r):.3f}; > 1.5: {(r > 1.5).sum()}, < 1/1.5: {(r < 1 / 1.5).sum()}; > 1.25: {(r > 1.25).sum()}, < 0.8: {(r < 0.8).sum()}")
    el = [x for x in rows if "elementwise" in x[3]]
    if el:
        re_ = np.array([x[0] for x in el])
        print(f"element-wise checks only ({len(el)}): geometric mean {np.exp(np.log(re_).mean()):.3f}; above the plain tolerance: {sum(x[1] > 1 for x in el)} ({args.label_a}) / {sum(x[2] > 1 for x in el)} ({args.label_b})")
    fmt = lambda x: f"{x[0]:8.2f}  {x[1]:10.3g}  {x[2]:10.3g}  {x[3][:120]}"  # noqa: E731
    print(f"\n   ratio  {args.label_a:>10s}  {args.label_b:>10s}  check   -- the {args.top} largest ratios")
    for x in rows[:args.top]:
        print(fmt(x))
    print(f"\n   -- the {args.top} smallest ratios")
    for x in rows[-args.top:]:
        print(fmt(x))
    if args.keys:
        print("\n   -- named checks")
        for x in rows:
            if any(s in x[3] for s in args.keys):
                print(fmt(x))


if __name__ == "__main__":
    main()
