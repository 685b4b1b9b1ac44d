"""Joins scripts/r04_pmc_dispatch.sh's output (gpurun_out/r04_pmcd.txt) into one table: algorithmic bytes of every conv
launch of a module_training step (from the library's launch trace) against the FETCH_SIZE / WRITE_SIZE of its dispatch."""
import re, sys
t = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04_pmcd.txt").read()
orders = re.findall(r"issue order \(dispatch index: MB\): (.*)", t)
f = [float(x.split(":")[1]) for x in orders[0].split()]
w = [float(x.split(":")[1]) for x in orders[1].split()]
rows = [l for l in t.splitlines() if re.match(r"\d+: conv_nhwc", l)]
tot = ta = 0.0
print("%3s %-18s %9s %9s %9s %6s" % ("", "site", "alg MB", "fetch MB", "write MB", "ratio"))
for i, l in enumerate(rows):
    parts = l.split()
    alg, site = float(parts[-1]), " ".join(parts[2:-4])
    tot += f[i] + w[i]
    ta += alg
    print("%3d %-18s %9.1f %9.1f %9.1f %6.2f" % (i, site, alg, f[i], w[i], (f[i] + w[i]) / alg))
print("step: %.2f GB measured, %.2f GB algorithmic, ratio %.3f" % (tot / 1e3, ta / 1e3, tot / ta))
