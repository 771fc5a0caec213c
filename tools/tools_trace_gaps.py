"""Where the time between kernels goes: reads a rocprofv3 kernel_trace.csv (--kernel-trace --output-format csv) of a bench.py run and
prints, for the last N steps (a step = the span from one `ncs_to_nsc_kernel` / `vprep_kernel` launch to the next), the wall time,
the sum of kernel durations, the idle time between consecutive kernels and the largest idle gaps with the kernels around them.

    python tools/tools_trace_gaps.py <kernel_trace.csv> [first-kernel-substring]
"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
first = sys.argv[2] if len(sys.argv) > 2 else "ncs_to_nsc_kernel"
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
starts = [i for i, k in enumerate(ks) if first in k[2]]
if len(starts) < 3:
    sys.exit("fewer than three steps in the trace")
steps = list(zip(starts[:-1], starts[1:]))[-5:]
short = lambda n: n.replace("msnet::", "").replace("(msnet::ConvArgs)", "")[:60]      # noqa: E731
tot_wall = tot_busy = 0.0
gaps = []
for a, b in steps:
    wall = ks[b][0] - ks[a][0]
    busy = 0
    end = ks[a][0]
    for i in range(a, b):
        s, e, n = ks[i]
        if s > end:
            gaps.append((s - end, short(ks[i - 1][2]) if i > a else "(step start)", short(n)))
        busy += max(0, e - max(s, end))
        end = max(end, e)
    if ks[b][0] > end:
        gaps.append((ks[b][0] - end, short(ks[b - 1][2]), "(next step) " + short(ks[b][2])))
    tot_wall += wall
    tot_busy += busy
n = len(steps)
print("steps %d: wall %.3f ms/step, some kernel running %.3f ms/step, idle %.3f ms/step (%d launches per step)"
      % (n, tot_wall / n / 1e6, tot_busy / n / 1e6, (tot_wall - tot_busy) / n / 1e6, steps[0][1] - steps[0][0]))
gaps.sort(reverse=True)
small = [g for g in gaps if g[0] < 20000]
print("idle gaps < 20 us: %d per step, mean %.1f us, total %.3f ms/step" % (len(small) / n, sum(g[0] for g in small) / max(1, len(small)) / 1e3,
                                                                            sum(g[0] for g in small) / n / 1e6))
print("largest gaps:")
for g in gaps[:12]:
    print("  %8.1f us  after %-60s before %s" % (g[0] / 1e3, g[1], g[2]))
