"""Per-layer HBM-side traffic of the conv launches of one forward: python scripts/traffic_per_layer.py <prof dir with fetch/ write/> <bench --breakdown log>
(the same pairing of dispatches and plan ops as scripts/traffic_from_pmc.py)"""
import csv, re, sys
prof, log = sys.argv[1:3]
ops = []
for line in open(log):
    m = re.match(r"\s+op\s+(\d+)\s+(conv|csp_fused)\s+([\d.]+) ms\s+[\d.]+ TF/s\s+[\d.]+ GB/s\s+([\d.]+) GF\s+([\d.]+) MB", line)
    if m:
        ops.append((int(m.group(1)), m.group(2), float(m.group(3)), float(m.group(4)), float(m.group(5))))
n = len(ops)
K = ("conv_igemm", "conv_patch", "conv_halo", "conv_pw", "conv_generic", "conv_stream", "csp_fused", "bneck_fused")
def per(path):
    rows = [r for r in csv.DictReader(open(path)) if any(k in r["Kernel_Name"] for k in K)]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows
f, w = per(prof + "/fetch/p_counter_collection.csv"), per(prof + "/write/p_counter_collection.csv")
assert len(f) % n == 0 and len(w) % n == 0, (len(f), len(w), n)
rf, rw = len(f) // n, len(w) // n
tot = 0.0
for k, (i, kind, ms, gf, mb) in enumerate(ops):
    fe = 2 * sum(float(r["Counter_Value"]) for r in f[k::n]) / rf * 1024 / 1e6
    wr = sum(float(r["Counter_Value"]) for r in w[k::n]) / rw * 1024 / 1e6
    name = re.sub(r".*::", "", f[k]["Kernel_Name"])[:40]
    tot += fe + wr - mb
    print(f"op{i:3d} {ms:.3f} ms {gf:7.2f} GF  alg {mb:7.2f} MB  read {fe:7.2f}  write {wr:7.2f}  x{(fe + wr) / mb:.2f}  excess {fe + wr - mb:7.1f}  {name}")
print(f"excess over algorithmic bytes, whole forward: {tot:.1f} MB")
