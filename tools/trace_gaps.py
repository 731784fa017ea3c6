import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last 30% of the trace (timed region)
t0 = int(rows[int(len(rows)*0.6)]["Start_Timestamp"])
rows = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
main = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k1_cols_fwd_c512","k2_rows_r16_planes","k3_cols_inv_c512","tail_cols","tail_rows"))]
span = int(main[-1]["End_Timestamp"]) - int(main[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main)
gaps = collections.defaultdict(list)
for a, b in zip(main, main[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    ka = a["Kernel_Name"].split("(")[0].split("::")[-1][:22]; kb = b["Kernel_Name"].split("(")[0].split("::")[-1][:22]
    gaps[(ka, kb)].append(g)
nh = sum(1 for r in main if "k3_cols_inv_c512" in r["Kernel_Name"])
print("haystacks", nh, "span/hay %.1f us  busy/hay %.1f us  gap/hay %.1f us" % (span/nh/1e3, busy/nh/1e3, (span-busy)/nh/1e3))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print("  %-24s -> %-24s n %4d  avg %7.2f us  total/hay %6.2f us" % (k[0], k[1], len(v), sum(v)/len(v)/1e3, sum(v)/nh/1e3))
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"].split("(")[0].split("::")[-1][:30]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("  %-32s n %4d avg %8.1f us" % (k, len(v), sum(v)/len(v)/1e3))
