"""Reads a rocprofv3 --kernel-trace CSV of tools/interactive_latency.py and prints, for the steady-state node-chain requests,
the GPU timeline: busy time, idle gaps and which kernels the long gaps precede.  Measurement aid only."""
import csv, glob, sys, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
print(len(rows), "kernel records")
# split into bursts separated by idle > 150 us (one burst ~ one request)
bursts, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > 150_000:
        bursts.append(cur); cur = [r]
    else:
        cur.append(r)
bursts.append(cur)
sizes = collections.Counter(len(b) for b in bursts)
print("burst sizes:", sizes.most_common(6))
n_chain = max(k for k, v in sizes.items() if v >= 10)
sel = [b for b in bursts if len(b) == n_chain][-20:]
span = sorted((b[-1][1] - b[0][0]) / 1e3 for b in sel)
busy = sorted(sum(e - s for s, e, _ in b) / 1e3 for b in sel)
print(f"{len(sel)} requests of {n_chain} kernels: span median {span[len(span)//2]:.1f} us, kernel-busy median {busy[len(busy)//2]:.1f} us")
gap_by = collections.defaultdict(list)
for b in sel:
    for (s0, e0, n0), (s1, e1, n1) in zip(b, b[1:]):
        gap_by[(n0[:40], n1[:40])].append((s1 - e0) / 1e3)
tot = collections.defaultdict(float)
for k, v in gap_by.items():
    tot[k] = sum(v) / len(sel)
print("largest idle gaps per request (us, summed over occurrences), prev kernel -> next kernel:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    g = gap_by[k]
    print(f"  {v:8.1f} us  ({len(g)//len(sel)} x {sum(g)/len(g):.1f})  {k[0]} -> {k[1]}")
dur = collections.defaultdict(list)
for b in sel:
    for s, e, n in b:
        dur[n[:60]].append((e - s) / 1e3)
print("kernel time per request (us):")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {sum(v)/len(sel):8.1f} us  ({len(v)//len(sel)} x {sum(v)/len(v):.1f})  {k}")
