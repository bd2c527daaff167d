"""diagnostic: timeline of a few rounds from a rocprofv3 --kernel-trace csv (start offsets, durations, stream / queue of every kernel)
   python scripts/round_timeline.py <dir with *_kernel_trace.csv> [first_fraction] [nkernels]"""
import csv, glob, os, sys
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
nk = int(sys.argv[3]) if len(sys.argv) > 3 else 70
fn = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = int(len(rows) * frac)
# start at an fft1w launch
while i0 < len(rows) and "k_fft1w" not in rows[i0]["Kernel_Name"]:
    i0 += 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + nk]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lrh::", "")
    print(f"{s / 1000:9.1f} {e / 1000:9.1f} {(e - s) / 1000:8.1f} us  q{r.get('Queue_Id', '?'):>3}  {name}")
