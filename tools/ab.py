#!/usr/bin/env python3
"""Tuning only: A/B timing of side builds of libdfe (tools/ubench/libdfe_<tag>.so) through bench.py.
Process-to-process placement noise on one box is ~+-5 %, so variants are interleaved over several rounds and the
minimum and median kernel_ms / ms_per_step are reported.   usage: ab.py ROUNDS tag [tag...]"""
import json, os, statistics, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, tags = int(sys.argv[1]), sys.argv[2:]
res = {t: [] for t in tags}
for r in range(rounds):
    for t in tags:
        lib, mode, tile = (t.split(":") + ["", ""])[:3]   # tag[:mode[:tile]] -- DFE_CV_MODE / DFE_CV_TILE
        env = dict(os.environ, DFE_LIB=os.path.join(root, "tools/ubench/libdfe_%s.so" % lib))
        if mode: env["DFE_CV_MODE"] = mode
        if tile: env["DFE_CV_TILE"] = tile
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "100"] + os.environ.get("AB_ARGS", "").split(), env=env,
                             capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(t, "FAILED", out.stderr[-500:]); sys.exit(1)
        j = json.loads(out.stdout.strip().splitlines()[-1])
        res[t].append((j["roofline"]["kernel_ms"], j["ms_per_step"], j.get("roofline_build_only", {}).get("kernel_ms", 0.0)))
for t in tags:
    k = sorted(x[0] for x in res[t]); s = sorted(x[1] for x in res[t])
    b = sorted(x[2] for x in res[t])
    print("%-10s kernel_ms min %.4f med %.4f | step min %.4f med %.4f | build-only min %.4f med %.4f" %
          (t, k[0], statistics.median(k), s[0], statistics.median(s), b[0], statistics.median(b)), flush=True)
