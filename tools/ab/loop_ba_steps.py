"""Per trial step of every BA window of a traced run (rocprofv3 --kernel-trace of tools/run_vo.py): microseconds by kernel.
Usage: python tools/ab/loop_ba_steps.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
starts = [i for i, r in enumerate(rows) if name(r) in ("ba_point_inverse_kernel", "ba_schur_strip_kernel", "ba_schur_kernel<true>", "ba_schur_kernel<false>")]
for a, b in zip(starts, starts[1:] + [None]):
    seg = rows[a:b] if b else rows[a:a + 60]
    tot, busy = {}, 0.0
    for r in seg:
        n = name(r)
        if not (n.startswith("ba_") or n.startswith("__amd_rocclr_fill")):
            break
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot[n] = tot.get(n, 0) + d
        busy += d
    if busy > 60:
        print(round(busy), {k.replace("ba_", "").replace("_kernel", ""): round(v, 1) for k, v in tot.items()})
