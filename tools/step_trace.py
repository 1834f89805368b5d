#!/usr/bin/env python3
"""The kernel SEQUENCE of one step from a rocprofv3 --kernel-trace csv: name, duration, gap to the previous kernel's end - where a step's
time goes beyond the big kernels (the plumbing launches and the idle gaps between them).

    python tools/step_trace.py gpurun_out/prof_c2l_<tag>_stats [steps_in_trace=7]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]      # the newest run (merged dirs keep old ones)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = len(rows) // n_steps
    # the last step: the trailing `per` kernels (setup kernels sit at the front of the trace)
    last = rows[-per:]
    t_end_prev = None
    tot = gap_tot = 0.0
    print("%d kernels in the trace, %d per step (last step shown)" % (len(rows), per))
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - t_end_prev) / 1e3 if t_end_prev is not None else 0.0
        dur = (e - s) / 1e3
        tot += dur
        gap_tot += max(gap, 0.0)
        print("%8.1f us  gap %7.1f  %s" % (dur, gap, r["Kernel_Name"][:110]))
        t_end_prev = e
    print("sum of kernels %.1f us, sum of gaps %.1f us, span %.1f us" % (tot, gap_tot, (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3))


if __name__ == "__main__":
    main()
