#!/usr/bin/env python3
"""The kernels around one step boundary of a rocprofv3 kernel trace (bench.py: 8 haystacks per am_match_batch_device call):
from the last K2 of a step to the first K2 of the next, start / end relative to that K2's end, in us."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.7):]
short = lambda r: r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
# a step boundary: the first tail_cols_fwd after a k3
idx = [i for i, r in enumerate(rows) if "tail_cols_fwd" in r["Kernel_Name"]]
i = idx[len(idx) // 2]
lo = max(j for j in range(i) if "k2_rows_r16_planes" in rows[j]["Kernel_Name"])
hi = min(j for j in range(i, len(rows)) if "k2_rows_r16_planes" in rows[j]["Kernel_Name"])
t0 = int(rows[lo]["End_Timestamp"])
for r in rows[lo:hi + 1]:
    print("%-30s start %8.1f  end %8.1f  (%6.1f us)  queue %s" % (short(r), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                           (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
