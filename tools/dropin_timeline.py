#!/usr/bin/env python3
"""Timeline of one T2 frame of the slot-shaped path from a rocprofv3 run of examples/t2gpu_rx_file (rocpd .db):
    T2GPU_DROPIN_WRAPPER="rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/dropin --" python bench.py --only-drop-in
    python tools/dropin_timeline.py gpurun_out/dropin
Prints, for the gap between the last equaliser launch of a frame and the first P1 correlator launch of the next, what the device did."""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
k = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
mc_tab = [t for t in tabs if "memory_cop" in t.lower() and "rocpd" not in t.lower()]
copies = []
if mc_tab:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % mc_tab[0])]
    name_col = "name" if "name" in cols else cols[0]
    size_col = "size" if "size" in cols else None
    copies = c.execute("select %s,start,end,%s from %s order by start" % (name_col, size_col or "0", mc_tab[0])).fetchall()
print("tables:", mc_tab, "kernels", len(k), "copies", len(copies))
p1 = [r for r in k if "p1_correlate" in r[0]]
# steady state: the last few P1 launches
for a, b in zip(p1[-4:-1], p1[-3:]):
    span = (b[1] - a[1]) / 1e6
    inside = [r for r in k if a[1] <= r[1] < b[1]]
    busy = sum(r[2] - r[1] for r in inside) / 1e6
    by = {}
    for r in inside:
        n = r[0].split("(")[0][-40:]
        by.setdefault(n, [0, 0.0])
        by[n][0] += 1; by[n][1] += (r[2] - r[1]) / 1e6
    print("frame: %.2f ms between P1 launches; %d kernels, sum of durations %.2f ms" % (span, len(inside), busy))
    for n, v in sorted(by.items(), key=lambda x: -x[1][1])[:14]:
        print("   %-42s %5d  %8.3f ms" % (n, v[0], v[1]))
    cin = [r for r in copies if a[1] <= r[1] < b[1]]
    byc = {}
    for r in cin:
        byc.setdefault(r[0], [0, 0.0, 0])
        byc[r[0]][0] += 1; byc[r[0]][1] += (r[2] - r[1]) / 1e6; byc[r[0]][2] += r[3] or 0
    for n, v in byc.items():
        print("   copy %-28s %5d  %8.3f ms  %10.1f MB" % (n, v[0], v[1], v[2] / 1e6))
    # the tail of the frame: from the last eq launch to the next P1 correlator
    eq = [r for r in inside if "eq_" in r[0] or "equal" in r[0]]
    if eq:
        t0 = eq[-1][2]
        print("   last equaliser end -> next P1 correlator start: %.3f ms" % ((b[1] - t0) / 1e6))
        for r in sorted([x for x in inside if x[1] >= t0] + [(("copy " + x[0]), x[1], x[2], -1) for x in cin if x[1] >= t0], key=lambda x: x[1])[:60]:
            print("      +%8.3f ms  %8.3f ms  %s" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[0].split("(")[0][-60:]))
    break

# ---- one data symbol in the middle of the last whole frame: everything between two FFT launches
fft = [r for r in k if "fft_fwd" in r[0] or "fft_stage_a" in r[0] or "fft_one_sync" in r[0] or "front_fft_one" in r[0]]   # a symbol's transform: its first (or only) launch
if len(fft) > 200:
    # how the symbol-to-symbol interval is distributed over the last frames (the one symbol printed below is a quiet one: intervals stretch
    # while SIMD batches of the frame before are being decoded on the same device)
    iv = sorted((b[1] - a[1]) / 1e3 for a, b in zip(fft[-181:-1], fft[-180:]))
    print("symbol-to-symbol interval over the last 180 symbols: min %.1f  p25 %.1f  median %.1f  p75 %.1f  p90 %.1f  max %.1f us, mean %.1f" % (
        iv[0], iv[len(iv) // 4], iv[len(iv) // 2], iv[3 * len(iv) // 4], iv[9 * len(iv) // 10], iv[-1], sum(iv) / len(iv)))
    ld = [r for r in k if "ldpc_decode" in r[0]]
    if ld:
        lo, hi = fft[-181][1], fft[-1][1]
        inside = [r for r in ld if r[2] > lo and r[1] < hi]
        print("LDPC launches overlapping that span: %d, %.2f ms each on average, their sum %.1f %% of the span" % (
            len(inside), sum(r[2] - r[1] for r in inside) / max(len(inside), 1) / 1e6, 100.0 * sum(min(r[2], hi) - max(r[1], lo) for r in inside) / (hi - lo)))
if len(fft) > 40:
    a, b = fft[-30], fft[-29]
    print("one symbol: %.1f us between FFT launches" % ((b[1] - a[1]) / 1e3))
    ops = [x for x in k if a[1] <= x[1] < b[1]] + [("copy " + x[0], x[1], x[2], -1) for x in copies if a[1] <= x[1] < b[1]]
    for r in sorted(ops, key=lambda x: x[1]):
        print("      +%8.1f us  %7.1f us  %s" % ((r[1] - a[1]) / 1e3, (r[2] - r[1]) / 1e3, r[0].split("(")[0][-60:] or "(unnamed kernel)"))

# ---- every symbol of the last whole frame: interval to the next FFT launch, and what else started on the device inside it
if len(p1) >= 3 and len(fft) > 130:
    a, b = p1[-3], p1[-2]
    fr = [r for r in fft if a[1] <= r[1] < b[1]]
    print("symbols of one frame (interval to the next symbol's FFT; kernels / copies other than the symbol chain's that START inside it):")
    chain = ("fft_stage", "fft_fwd", "fft_one_sync", "front_fft_one", "sym_sync", "eq_split", "publish_symbol", "ti_scatter", "front_", "cp_correlate", "eq_sync")
    for x, y in zip(fr[:-1], fr[1:]):
        ins = [r for r in k if x[1] <= r[1] < y[1] and r[0] and not any(c in r[0] for c in chain)]
        cps = [r for r in copies if x[1] <= r[1] < y[1]]
        un = [r for r in k if x[1] <= r[1] < y[1] and not r[0]]
        txt = ", ".join("%s %.0f" % (r[0].split("(")[0].split("::")[-1][:22], (r[2] - r[1]) / 1e3) for r in ins[:6])
        txt += "".join(", copy %.0f (%.1f MB)" % ((r[2] - r[1]) / 1e3, (r[3] or 0) / 1e6) for r in cps[:4])
        print("   %7.1f us  front %s  %s" % ((y[1] - x[1]) / 1e3, "+".join("%.0f" % ((r[2] - r[1]) / 1e3) for r in un), txt))
    # the slowest symbols of that frame in full: every kernel / copy that starts inside, with its stream
    slow = sorted(zip(fr[:-1], fr[1:]), key=lambda xy: xy[0][1] - xy[1][1])[:3]
    for x, y in slow:
        print("   --- a %.1f us symbol:" % ((y[1] - x[1]) / 1e3))
        ops = [(r[0], r[1], r[2], r[3]) for r in k if r[1] < y[1] and r[2] > x[1]] + [("copy " + r[0], r[1], r[2], -1) for r in copies if r[1] < y[1] and r[2] > x[1]]
        for r in sorted(ops, key=lambda q: q[1]):
            print("      +%8.1f us  %7.1f us  stream %3s  %s" % ((r[1] - x[1]) / 1e3, (r[2] - r[1]) / 1e3, r[3], (r[0] or "(unnamed)").split("(")[0][-50:]))
