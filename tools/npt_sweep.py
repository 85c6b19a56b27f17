"""Sweep SPGNN_NPT (nodes per team) for the GAT kernels: one subprocess per value (the env var is read once)."""
import os, subprocess, sys
for npt in [1, 2, 4, 8, 16]:
    env = dict(os.environ, SPGNN_NPT=str(npt))
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "microbench.py"), "--trees", "512", "--out", f"/tmp/mb{npt}.json"],
                         env=env, capture_output=True, text=True).stdout
    rows = [l for l in out.splitlines() if l.startswith("{'H'")]
    tot_f = tot_b = 0
    import ast
    for l in rows:
        r = ast.literal_eval(l); tot_f += r["fwd_ms"]; tot_b += r["bwd_ms"]
    print(f"npt={npt}: fwd total {tot_f*1e3:.0f} us, bwd total {tot_b*1e3:.0f} us | " + " ".join(f"({ast.literal_eval(l)['H']},{ast.literal_eval(l)['D']}):{ast.literal_eval(l)['fwd_ms']*1e3:.0f}/{ast.literal_eval(l)['bwd_ms']*1e3:.0f}" for l in rows), flush=True)
