#!/usr/bin/env python3
"""Run bench.py with the given arguments and print a one-line summary (tuning helper)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-4000:])
    sys.exit(1)
d = json.loads(line[-1])
tag = os.environ.get("TAG", "")
msg = f"{tag} value={d['value']:.1f} {d['unit']} ms_per_step={d['ms_per_step']:.2f} sweeps={d['config'].get('jacobi_sweeps_max')}"
if "roofline" in d:
    r = d["roofline"]
    msg += f" | large mesh: {r['fct_steps_per_s']:.1f} steps/s, {r['jacobi_sweeps_per_step']:.0f} sweeps/step; " + " ".join(
        f"{k}={v['achieved_GBps']:.0f}GB/s({v['avg_launch_ms'] * 1e3:.0f}us x{v['launches']})" for k, v in r["kernels"].items())
if "pgd_c2" in d:
    g = d["pgd_c2"]
    msg += f" | PGD: spec {g['speculative']['s_per_pgd_iteration']*1e3:.1f} ms/iter, seq {g['sequential']['s_per_pgd_iteration']*1e3:.1f} ms/iter, trials {g['sequential']['armijo_trials']}, dJ={g.get('cost_rel_diff')}"
print(msg)
