"""Occupancy account of ONE k_vocoder_lt launch from per-wave stamps (tools/lt_occupancy.sh): for every SIMD the time
it held two, one and no waves of the kernel between the kernel's first wave start and last wave end."""
import collections
import sys

rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if l.strip() and not l.startswith("#")]
rows = [r for r in rows if r[1] and r[2]]
t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
span = (t1 - t0) / 1e5  # ms (100 MHz clock)
simd = collections.defaultdict(list)
for w, s, e, hw, xcc, fr in rows:
    # HW_ID: SIMD_ID bits 5:4, CU_ID 11:8, SH_ID 12, SE_ID 15:13 (gfx9)
    key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)
    simd[key].append((s, e, fr))
starts = sorted((r[1] - t0) / 1e5 for r in rows)
ends = sorted((r[2] - t0) / 1e5 for r in rows)
dur = sorted((r[2] - r[1]) / 1e5 for r in rows)
q = lambda a, p: a[min(len(a) - 1, int(p * len(a)))]
print(f"{len(rows)} waves on {len(simd)} SIMDs; first start -> last end {span:.2f} ms")
print(f"wave start after the first:  median {q(starts, .5):.3f}  90 % {q(starts, .9):.3f}  99 % {q(starts, .99):.3f}  last {starts[-1]:.3f} ms")
print(f"wave end before the last:    first {span - ends[0]:.3f}  1 % {span - q(ends, .01):.3f}  10 % {span - q(ends, .1):.3f}  median {span - q(ends, .5):.3f} ms")
print(f"wave duration:               min {dur[0]:.2f}  median {q(dur, .5):.2f}  max {dur[-1]:.2f} ms")
acc = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
per = collections.Counter(len(v) for v in simd.values())
for key, ws in simd.items():
    ev = sorted([(s, 1) for s, e, f in ws] + [(e, -1) for s, e, f in ws])
    n, prev = 0, t0
    for t, d in ev:
        acc[min(n, 3)] += (t - prev) / 1e5
        n, prev = n + d, t
    acc[min(n, 3)] += (t1 - prev) / 1e5
ns = len(simd)
print(f"waves per SIMD: {dict(sorted(per.items()))}")
print(f"mean over SIMDs of the time holding 0 / 1 / 2 / 3+ waves: {acc[0] / ns:.2f} / {acc[1] / ns:.2f} / {acc[2] / ns:.2f} / {acc[3] / ns:.2f} ms of {span:.2f}")
print(f"SQ_WAVE_CYCLES / 2 equivalent (wave-time / 2 per SIMD): {(acc[1] + 2 * acc[2] + 3 * acc[3]) / ns / 2:.2f} ms")
# the under-occupied time by cause: ramp (before a SIMD's second wave starts), tail (after its first wave ends)
ramp = tail = 0.0
for ws in simd.values():
    if len(ws) >= 2:
        s = sorted(w[0] for w in ws)
        e = sorted(w[1] for w in ws)
        ramp += (s[1] - t0) / 1e5
        tail += (t1 - e[-2]) / 1e5
    else:
        ramp += span
print(f"per SIMD, mean: until its second wave has started {ramp / ns:.2f} ms; after its second-to-last wave has ended {tail / ns:.2f} ms")
fr = sorted(r[5] for r in rows)
print(f"frames per wave (chunk + warm-up): min {fr[0]} median {q(fr, .5)} max {fr[-1]}")
byx = collections.defaultdict(list)
for r in rows:
    byx[r[4]].append((r[2] - t0) / 1e5)
print("last wave end per XCD:", {x: round(max(v), 2) for x, v in sorted(byx.items())})
