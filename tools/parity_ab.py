"""Error STATISTICS of the three arrays VERDICT r4 named (golden realnvp_d64_h64 `ys`, cfg 5 `ladj_inv`, cfg 4 `ladj`) for
whatever arithmetic the environment selects -- run once as is (six-term bf16 products, the default) and once with
NF_FWD_FP32=1 NF_BWD_FP32=1 NF_WIDE_FP32=1 (fp32 MFMA chains), on one box; tests/test_gpu_tape.py does exactly that and
compares.  Why statistics: the recorded parity figure of such an array is the MAX over 256 sampled columns of
|err| / (atol + rtol |ref|), i.e. the one worst-conditioned sample of a deep flow times whatever the roundings project on
it -- two arithmetics of the same quality land a factor 2-3 apart on it by chance (the float32 numpy oracle itself sits at
7.8 x on cfg 5's ladj_inv).  A one-sided arithmetic error -- what round 4's truncating split had -- shows in the MEAN
signed error and in the RMS over many columns, which is what this script reports next to the max (4 096 columns for cfg 5,
1 024 for cfg 4, the golden's own 64 x 96 elements).

usage: python3 tools/parity_ab.py out.json          (needs a GPU; the oracle is the checker)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stats(got, ref, floor):
    got, ref, floor = (np.asarray(a, dtype=np.float64) for a in (got, ref, floor))
    tol = 1e-6 + 1e-5 * np.abs(ref)
    e, f = (got - ref) / tol, (floor - ref) / tol
    q = lambda v: {"max_x_tol": float(np.abs(v).max()), "rms_x_tol": float(np.sqrt((v * v).mean())), "mean_signed_x_tol": float(v.mean())}  # noqa: E731
    return {"device": q(e), "float32_numpy_oracle": q(f), "n": int(e.size)}


def main(out):
    import torch

    import nf_oracle as o
    from __graft_entry__ import load_package
    import parity as P

    nf = load_package()
    res = {"env": {k: os.environ.get(k, "") for k in ("NF_FWD_FP32", "NF_BWD_FP32", "NF_WIDE_FP32")}}
    # (1) golden realnvp_d64_h64: ys of the committed xs
    z = np.load(os.path.join(ROOT, "tests", "golden", "realnvp_d64_h64.npz"), allow_pickle=True)
    d, nl, hd = int(z["d"]), int(z["nlayers"]), tuple(int(h) for h in z["hdims"])
    flow = nf.Flow("realnvp", nf.MvNormal(d), nl, hd, 0, 0.0, dtype=torch.float32, device="cuda",
                   theta=torch.tensor(z["theta"], dtype=torch.float32, device="cuda"))
    spec = o.FlowSpec("realnvp", d, nl, hd)
    xs = torch.tensor(np.ascontiguousarray(z["xs"].T), dtype=torch.float32, device="cuda").T
    ys, ladj = nf.with_logabsdet_jacobian(flow.transform, xs)
    y32, l32 = o.flow_fwd(spec, *P.f32(z["theta"], z["xs"]))
    res["golden realnvp_d64_h64: ys"] = stats(ys.cpu().numpy(), z["ys"], y32)
    res["golden realnvp_d64_h64: ladj"] = stats(ladj.cpu().numpy(), z["ladj"], l32)
    # (2) cfg 5: ladj of the inverse chain, 4 096 sampled columns of the 1 M-sample batch
    d, n = 64, 1 << 20
    flow = nf.realnvp(nf.MvNormal(d), (64, 64), 4, paramtype=torch.float32, seed=123)
    gen = torch.Generator().manual_seed(5)
    flow = flow.with_theta(flow.theta + 0.02 * torch.randn(flow.P, generator=gen).to("cuda"))
    spec = o.FlowSpec("realnvp", d, 4, (64, 64))
    th64 = flow.theta.cpu().numpy().astype(np.float64)
    ys = nf.device_specific_rand(nf.PhiloxRNG(123), flow.dist, n) * 2 + 1
    cols = np.sort(np.random.default_rng(9).choice(n, 4096, replace=False))
    ci = torch.tensor(cols, device="cuda")
    y_sel = ys[:, ci].cpu().numpy().astype(np.float64)
    xr, ladj = nf.with_logabsdet_jacobian(nf.inverse(flow.transform), ys)
    x_ref, l_ref = o.flow_inv(spec, th64, y_sel)
    x32, l32 = o.flow_inv(spec, *P.f32(th64, y_sel))
    res["cfg5: ladj_inv (4096 sampled columns)"] = stats(ladj[ci].cpu().numpy(), l_ref, l32)
    res["cfg5: x = T^-1 y (4096 sampled columns)"] = stats(xr[:, ci].cpu().numpy(), x_ref, x32)
    # (3) cfg 4 (damped initialisation, the parity suite's): ladj of 1 024 sampled columns of rank 3's shard
    d, nl, hd, n, off = 256, 8, (256, 256), 32768, 3 * 32768
    flow = nf.realnvp(nf.MvNormal(d), hd, nl, paramtype=torch.float32, seed=123)
    gen = torch.Generator().manual_seed(5)
    flow = flow.with_theta(0.5 * (flow.theta + 0.02 * torch.randn(flow.P, generator=gen).to("cuda")))
    spec = o.FlowSpec("realnvp", d, nl, hd)
    th64 = flow.theta.cpu().numpy().astype(np.float64)
    xs = nf.device_specific_rand(nf.PhiloxRNG(123, sample_offset=off), flow.dist, n)
    cols = np.sort(np.random.default_rng(7).choice(n, 1024, replace=False))
    ci = torch.tensor(cols, device="cuda")
    x_sel = xs[:, ci].cpu().numpy().astype(np.float64)
    ys, ladj = nf.with_logabsdet_jacobian(flow.transform, xs)
    y_ref, l_ref = o.flow_fwd(spec, th64, x_sel)
    y32, l32 = o.flow_fwd(spec, *P.f32(th64, x_sel))
    res["cfg4: ladj (1024 sampled columns)"] = stats(ladj[ci].cpu().numpy(), l_ref, l32)
    res["cfg4: ys (1024 sampled columns)"] = stats(ys[:, ci].cpu().numpy(), y_ref, y32)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        if k != "env":
            print(f"{k:45s} device max {v['device']['max_x_tol']:7.3f} rms {v['device']['rms_x_tol']:7.4f} mean {v['device']['mean_signed_x_tol']:+8.4f}"
                  f" | numpy f32 max {v['float32_numpy_oracle']['max_x_tol']:7.3f} rms {v['float32_numpy_oracle']['rms_x_tol']:7.4f} mean {v['float32_numpy_oracle']['mean_signed_x_tol']:+8.4f}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_ab.json")
