"""Per-launch durations of the level-scheduled factorisation from a rocprofv3 --kernel-trace csv (mean over the solves of the trace)."""
import csv, sys, glob, collections
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
seq, cur = [], []
for r in rows:
    k = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_chol_level" in k:
        cur.append((d, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))))
    elif cur:
        seq.append(cur); cur = []
n = max(len(s) for s in seq)
seq = [s for s in seq if len(s) == n][2:]
print("solves", len(seq), "launches", n)
for l in range(n):
    dur = sum(s[l][0] for s in seq) / len(seq)
    gap = sum((s[l][1] - s[l - 1][2]) / 1e3 for s in seq) / len(seq) if l else 0.0
    print(f"launch {l:2d}: grid {seq[0][l][3]:7d}  {dur:7.2f} us  gap before {gap:5.2f} us")
print("chain", sum((s[-1][2] - s[0][1]) / 1e3 for s in seq) / len(seq), "us")
