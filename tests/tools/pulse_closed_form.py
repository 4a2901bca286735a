"""The pulse walk in closed form, bit for bit (CPU model of k_pulse_queue's fast frame; round 4).

Excitation::get's voiced branch (src/vocoder/excitation.rs:73-81) runs, per sample,
    counter += 1; fire = counter >= cur; if fire: counter -= cur; cur += inc
and only `counter` is carried from frame to frame.  Two facts make a frame O(fires) instead of O(samples):
  * while cur stays in one binade, fl(cur + inc) = cur + d with d = RN_ulp(inc) (no tie): cur_j = cur_0 + j*d
    exactly;
  * n increments of the counter are exact whenever c + n is representable (all c + k, k <= n, then are: they
    share c's fractional bits and are no larger) -- one TwoSum tells; otherwise the walk stays below the top
    of the counter's binade and takes the crossing as one rounded addition, as the loop does.
The first fire of a stretch is estimated by a division and settled by evaluating both (exact) sides.
Round 2's event-driven form (tools/experiments/pulse_event_driven.py) stopped at EVERY binade top of the
counter (~9 per pitch period); here the exactness test makes the common stretch the whole rest of the frame.
This script checks the model against the per-sample loop on random frames: fires and final counter, bit for bit.
"""
import math
import random
import struct


def bits(x): return struct.unpack('<q', struct.pack('<d', x))[0]
def frombits(b): return struct.unpack('<d', struct.pack('<q', b))[0]
def expo(x): return (bits(x) >> 52) & 0x7ff
def top_of(x): return frombits((expo(x) + 1) << 52)
def bot_of(x): return frombits((expo(x)) << 52)
def ulp_of(x): return frombits((expo(x) - 52) << 52)


def brute(c, u, inc, fp):
    fires = []
    for j in range(fp):
        c = c + 1.0
        if c >= u:
            fires.append(j); c = c - u
        u = u + inc
    return fires, c


def two_sum_exact(a, b):
    s = a + b
    bb = s - a
    err = (a - (s - bb)) + (b - bb)
    return err == 0.0


def fast_frame(c, u0, inc, fp, stats):
    """fires and final counter of one frame.  cur runs through SEGMENTS, each inside one binade (a frame has one,
    or two when the pitch crosses a power of two); a sample whose cur cannot start a segment (tie increment,
    |d| >= 1, denormal range) is taken the loop's way, one at a time."""
    fires = []
    j = 0
    us = u0                      # cur at sample j (exact)
    while j < fp:
        # ---- set up the cur-segment that starts at sample j: cur_{j+k} = us + k*d for k = 0..kmax ----
        ok = us > 0.0 and expo(us) > 60
        if ok:
            u1 = us + inc
            d = u1 - us
            U = ulp_of(us)
            r = inc - d
            # no tie -- or a tie from an even mantissa: round-to-even then lands on even mantissas for good
            # (increments of (p - p')/fperiod are ties once in fperiod frames: p - p' is a multiple of the ulp)
            ok = (expo(u1) == expo(us) and -8.0 < d < 0.9 and
                  (abs(r) < 0.5 * U or (abs(r) == 0.5 * U and (bits(us) & 1) == 0)))
        if not ok:
            stats['slowstep'] += 1
            c = c + 1.0
            if c >= us:
                fires.append(j); c = c - us
            us = us + inc
            j += 1
            continue
        rem = fp - 1 - j
        if d > 0.0:
            room = top_of(us) - us
            q = room / d
            k = rem if q > rem + 2 else min(max(int(math.ceil(q)) - 1, 0), rem)
            while k > 0 and not (us + k * d < top_of(us)): k -= 1
            while k < rem and (us + (k + 1) * d < top_of(us)): k += 1
        elif d < 0.0:
            room = us - bot_of(us)
            q = room / -d
            k = rem if q > rem + 2 else min(max(int(math.floor(q)), 0), rem)
            while k > 0 and not (us + k * d >= bot_of(us)): k -= 1
            while k < rem and (us + (k + 1) * d >= bot_of(us)): k += 1
        else:
            k = rem
        jend = j + k             # last sample of the segment
        js = j
        # ---- the counter through the segment, a stretch of exact increments at a time ----
        while j <= jend:
            stats['iters'] += 1
            lim = jend - j + 1
            uj = us + (j - js) * d
            if two_sum_exact(c, float(lim)):
                M = lim
            elif c >= 1.0:
                M = min(int(math.ceil(top_of(c) - c)) - 1, lim)
            else:
                M = 0
            if M <= 0:
                c = c + 1.0
                if c >= uj:
                    fires.append(j); c = c - uj
                j += 1
                continue
            def cond(m): return (c + (m + 1)) >= (uj + m * d)
            a = uj - c - 1.0
            if a <= 0.0:
                m = 0
            else:
                est = math.ceil(a / (1.0 - d))
                m = int(est) if est < M else M
            while m > 0 and cond(m - 1): m -= 1; stats['adj'] += 1
            while m < M and not cond(m): m += 1; stats['adj'] += 1
            if m < M:
                c = (c + (m + 1)) - (uj + m * d)
                fires.append(j + m)
                j += m + 1
            else:
                c = c + M
                j += M
        # leave the segment: the update that crosses the binade is a rounded addition
        us = (us + (jend - js) * d) + inc
    return fires, c


if __name__ == "__main__":
    random.seed(3)
    stats = {'iters': 0, 'adj': 0, 'slowstep': 0}
    nbad = tot = 0
    N = 300000
    for it in range(N):
        mode = random.random()
        if mode < 0.5:
            u0 = random.uniform(100, 700); inc = random.uniform(-5, 5) / 240
        elif mode < 0.65:
            u0 = random.uniform(2.4, 40); inc = random.uniform(-2, 2) / 240
        elif mode < 0.8:
            u0 = random.uniform(2.4, 2400); inc = random.uniform(-u0, 2400 - u0) / 240
        else:
            u0 = float(random.choice([128, 256, 512, 64, 4, 8])) + random.uniform(-1, 1); inc = random.uniform(-3, 3) / 240
        if random.random() < 0.2: inc = 0.0
        if random.random() < 0.05: inc = float(random.choice([0.5, 0.25, -0.125, 1 / 1024, 3 / 4096]))
        if random.random() < 0.1: inc = (random.randint(-2000, 2000) + 0.5) * ulp_of(u0)  # exact ties
        cm = random.random()
        c = u0 if cm < 0.1 else random.uniform(0, u0 + 1)
        if cm > 0.95: c = random.random() * 1e-9
        fp = random.choice([240, 80, 90, 100])
        f1, c1 = brute(c, u0, inc, fp)
        tot += fp
        f2, c2 = fast_frame(c, u0, inc, fp, stats)
        if f1 != f2 or bits(c1) != bits(c2):
            nbad += 1
            if nbad < 5:
                print("MISMATCH", repr(c), repr(u0), repr(inc), fp, f1[:5], f2[:5], c1, c2)
    print("frames", N, "bad", nbad, "samples taken the loop's way", stats['slowstep'], "of", tot, "iterations per frame",
          stats['iters'] / N, "adjust steps per frame", stats['adj'] / N)
