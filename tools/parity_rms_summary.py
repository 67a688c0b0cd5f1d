"""Device against IEEE float32 (the oracle evaluated op by op in numpy float32) over every element-wise check of a GPU run's parity table
that carries a float32 floor: per check the MAXIMUM and the RMS of error / (atol + rtol |ref|) for both, and the summary statistics of
the two ratios.  usage: python tools/parity_rms_summary.py gpurun_out/parity_measured.json"""
import json
import sys

import numpy as np

t = json.load(open(sys.argv[1]))
M, FM = " [elementwise err / (atol + rtol|ref|)]", " [fp32-oracle floor / (atol + rtol|ref|)]"
R, FR = " [rms of elementwise err / (atol + rtol|ref|)]", " [fp32-oracle floor, rms / (atol + rtol|ref|)]"
rows = []
for k in t:
    if k.endswith(FR):
        b = k[: -len(FR)]
        if b + R in t and b + M in t and b + FM in t and t[k] > 0 and t[b + FM] > 0:
            rows.append((t[b + R] / t[k], t[b + M] / t[b + FM], t[b + R], t[k], t[b + M], t[b + FM], b))
rows.sort(reverse=True)
rr, mr = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
print(f"{len(rows)} element-wise checks with a float32 floor")
print(f"device rms / float32-oracle rms: geometric mean {np.exp(np.log(rr).mean()):.3f}, median {np.median(rr):.3f}, max {rr.max():.2f}, "
      f"> 1.5: {(rr > 1.5).sum()}, > 2: {(rr > 2).sum()}")
print(f"device max / float32-oracle max: geometric mean {np.exp(np.log(mr).mean()):.3f}, median {np.median(mr):.3f}, max {mr.max():.2f}, "
      f"> 1.5: {(mr > 1.5).sum()}, > 2: {(mr > 2).sum()}")
print(f"device rms above the plain tolerance: {(np.array([r[2] for r in rows]) > 1).sum()}; float32 oracle: {(np.array([r[3] for r in rows]) > 1).sum()}")
print("\n rms ratio  max ratio |  dev rms  f32 rms |  dev max  f32 max | check   -- the 25 largest rms ratios")
for r in rows[:25]:
    print(f"   {r[0]:6.2f}    {r[1]:6.2f}  | {r[2]:8.3g} {r[3]:8.3g} | {r[4]:8.3g} {r[5]:8.3g} | {r[6][:90]}")
