"""Kernel timeline of the LAST evaluation in a rocprofv3 --kernel-trace directory: tools/timeline.py DIR [min_ms]
(start and end in ms relative to the first kernel of the evaluation, stream, kernel name)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void tehmm::", ""),
                     r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
minms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
# the last evaluation starts at the last k_emis_gain_lane / k_fused_rowindex ... take the last gap > 20 ms or the last "k_repack"/first kernel
starts = [i for i, r in enumerate(rows) if "k_emis_gain_lane" in r[2] or "k_wide_logrows" in r[2]]
i0 = starts[-1] if starts else 0
# include kernels of the same evaluation that started slightly earlier (posterior stream): back up to 2 ms
t0 = rows[i0][0]
j = i0
while j > 0 and rows[j - 1][0] > t0 - 3_000_000:
    j -= 1
t0 = rows[j][0]
for s, e, n, q in rows[j:]:
    if (e - s) / 1e6 >= minms:
        print("%8.2f %8.2f  %6.2f ms  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n[:70]))
