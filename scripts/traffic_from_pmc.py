"""HBM traffic per conv launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py, split at the ridge
point like bench.py's roofline.  Usage:
    python scripts/traffic_from_pmc.py <prof dir with fetch/ and write/> <bench --breakdown log> <out.json>
The k-th conv_igemm dispatch of a forward is the k-th conv op of the plan (same order in every forward).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests of 16-B-per-lane loads as 64 B -> x2;
WRITE_SIZE is exact.  Both counters are in KiB."""
import collections, csv, json, re, sys

prof, log, out = sys.argv[1:4]
ops = []
op_ms = []
for line in open(log):
    m = re.match(r"\s+op\s+(\d+)\s+(conv|csp_fused)\s+([\d.]+) ms\s+[\d.]+ TF/s\s+[\d.]+ GB/s\s+([\d.]+) GF\s+([\d.]+) MB", line)
    if m:
        # a fused CSP launch is priced against HBM like bench.py does: its flops are zeroed for the split at the ridge
        ops.append((0.0 if m.group(2) == "csp_fused" else float(m.group(4)) * 1e9, float(m.group(5)) * 1e6))
        op_ms.append(float(m.group(3)))
n = len(ops)

def per_dispatch(path):
    rows = [r for r in csv.DictReader(open(path)) if any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv_patch", "conv_halo", "conv_pw", "conv_generic", "conv_stream", "csp_fused", "bneck_fused"))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) * 1024.0 for r in rows]

fetch = per_dispatch(prof + "/fetch/p_counter_collection.csv")
write = per_dispatch(prof + "/write/p_counter_collection.csv")
assert len(fetch) % n == 0 and len(write) == len(fetch), (len(fetch), len(write), n)
reps = len(fetch) // n
ridge = 2500e12 / 8000e9
res = {"hot": collections.defaultdict(float), "thin": collections.defaultdict(float)}
for k, (fl, by) in enumerate(ops):
    g = res["hot" if fl / by >= ridge else "thin"]
    g["n"] += 1
    g["alg_bytes"] += by
    g["fetch_x2"] += 2.0 * sum(fetch[k::n]) / reps
    g["write"] += sum(write[k::n]) / reps
o = {"source": prof, "forwards_averaged": reps, "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950, KiB)"}
for name, g in res.items():
    o[name] = {"launches": int(g["n"]), "algorithmic_bytes_per_launch": g["alg_bytes"] / g["n"],
               "hbm_bytes_per_launch": (g["fetch_x2"] + g["write"]) / g["n"],
               "read_bytes_per_launch": g["fetch_x2"] / g["n"], "write_bytes_per_launch": g["write"] / g["n"]}
# the same grouping applied to the one-stream kernel trace (rocprofv3 --kernel-trace, CCNET_LANES=0), when present: the average
# launch duration bench.py's live HIP-event timing of this group has to agree with
import os
tr = prof + "/trace_serial/t_kernel_trace.csv"
if os.path.exists(tr):
    rows = [r for r in csv.DictReader(open(tr)) if any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv_patch", "conv_halo", "conv_pw", "conv_generic", "conv_stream", "csp_fused", "bneck_fused"))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    if len(dur) % n == 0:
        rr = len(dur) // n
        for name in ("hot", "thin"):
            ks = [k for k, (fl, by) in enumerate(ops) if (fl / by >= ridge) == (name == "hot")]
            o[name]["trace_avg_launch_us"] = sum(sum(dur[k::n]) / rr for k in ks) / len(ks)
            o[name]["bench_event_avg_launch_us"] = sum(ms for k, ms in enumerate(op_ms) if k in set(ks)) * 1e3 / len(ks)
        o["trace_forwards_averaged"] = rr
json.dump(o, open(out, "w"), indent=1)
print(json.dumps(o, indent=1))
