# developer tool: gaps between consecutive k_step dispatches in a rocprofv3 kernel trace (usage: kernel_gaps.py <dir>)
import csv, glob, sys
import numpy as np
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=lambda p: __import__("os").path.getmtime(p))
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps = []; durs = []
for a, b in zip(rows, rows[1:]):
    if "k_step" in a["Kernel_Name"] and "k_step" in b["Kernel_Name"]:
        gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
        durs.append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
g = np.array(gaps); d = np.array(durs)
print(f, len(g), "gaps: median %.0f ns mean %.0f p10 %.0f p90 %.0f ; k_step dur median %.0f mean %.0f" % (np.median(g), g.mean(), np.percentile(g, 10), np.percentile(g, 90), np.median(d), d.mean()))
