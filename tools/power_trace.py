"""Socket power and shader clock while bench.py runs (GPU box): samples the hwmon files of the device every few ms in
a thread (amdgpu: power1_average / power1_input in microwatts, freq1_input in Hz) and prints the distribution.
Usage: python tools/power_trace.py [bench.py arguments]"""
import glob, subprocess, sys, threading, time, statistics as st

def cards():
    """hwmon directories of every amdgpu device the box shows (all GPUs of the host appear in sysfs, one is ours)"""
    out = []
    for h in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
        d = {}
        for name in ('power1_average', 'power1_input', 'freq1_input', 'power1_cap'):
            for p in glob.glob(h + '/' + name):
                d[name] = p
        if 'power1_average' in d or 'power1_input' in d:
            out.append(d)
    return out

def rd(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None

C = cards()
allsamples, stop = [[] for _ in C], False
def loop():
    while not stop:
        for F, sm in zip(C, allsamples):
            sm.append((time.time(), rd(F.get('power1_average') or F.get('power1_input', '')), rd(F.get('freq1_input', ''))))
        time.sleep(0.004)
t = threading.Thread(target=loop); t.start()
args = sys.argv[1:] or ['--no-cpu-baseline', '--no-extras', '--steps', '30', '--warmup', '3']
t0 = time.time()
r = subprocess.run([sys.executable, 'bench.py'] + args, capture_output=True, text=True)
t1 = time.time()
stop = True; t.join()
print(r.stdout.strip().splitlines()[-1][:300] if r.stdout.strip() else r.stderr[-500:])
# ours is the device whose power moved most during the run
spread = [max([s[1] or 0 for s in sm] or [0]) - min([s[1] or 0 for s in sm] or [0]) for sm in allsamples]
k = spread.index(max(spread)); F, samples = C[k], allsamples[k]
print('device', k, 'of', len(C), F.get('power1_input') or F.get('power1_average'))
print('power cap (W):', (rd(F['power1_cap']) or 0) / 1e6 if 'power1_cap' in F else None)
# the timed steps are the last part of the run: take the last 30 % of the samples
n = len(samples); tail = samples[int(n * 0.7):]
pw = [s[1] / 1e6 for s in tail if s[1]]; fq = [s[2] / 1e6 for s in tail if s[2]]
if pw: print('power W (last 30 %% of the run, %d samples): min %.0f median %.0f max %.0f' % (len(pw), min(pw), st.median(pw), max(pw)))
if fq: print('sclk MHz: min %.0f median %.0f max %.0f' % (min(fq), st.median(fq), max(fq)))
pw_all = [s[1] / 1e6 for s in samples if s[1]]
if pw_all: print('power W over the whole run: max %.0f' % max(pw_all))
