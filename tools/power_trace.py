"""Socket power and shader clock of THE CARD UNDER TEST while bench.py runs (GPU box).  Every GPU of the host shows
in sysfs and one is ours: the card is found by the PCI bus id HIP reports for device 0 (asked in a child process, so
that the sampler itself never initialises the GPU), not by guessing from the readings (round 4 read a neighbour).
Samples the card's hwmon files every few ms in a thread (amdgpu: power1_average / power1_input in microwatts,
freq1_input = sclk in Hz) and prints the distribution over the timed steps.
Usage: python tools/power_trace.py [bench.py arguments]"""
import glob, os, subprocess, sys, threading, time, statistics as st

def our_bus_id():
    code = ("import ctypes;h=ctypes.CDLL('libamdhip64.so');b=ctypes.create_string_buffer(64);"
            "rc=h.hipDeviceGetPCIBusId(b,64,0);print(b.value.decode() if rc==0 else '')")
    try:
        return subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120).stdout.strip().lower()
    except Exception:
        return ''

def hwmon_of(bus):
    d = {}
    for h in sorted(glob.glob(f'/sys/bus/pci/devices/{bus}/hwmon/hwmon*')):
        for name in ('power1_average', 'power1_input', 'freq1_input', 'freq2_input', 'power1_cap', 'in0_input'):
            for p in glob.glob(h + '/' + name):
                d[name] = p
        for p in glob.glob(h + '/temp*_input'):  # edge / junction (hotspot) / memory, by their labels
            try:
                lab = open(p.replace('_input', '_label')).read().strip()
            except Exception:
                lab = os.path.basename(p)
            d['temp_' + lab] = p
            for lim in ('_crit', '_emergency', '_max'):
                q = p.replace('_input', lim)
                if os.path.exists(q):
                    d['limit_' + lab + lim] = q
    return d

def rd(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None

bus = our_bus_id()
F = hwmon_of(bus) if bus else {}
print('device 0 PCI bus id:', bus or 'unknown', '| hwmon files:', {k: v for k, v in F.items()})
if not F:
    print('no hwmon directory for that bus id is visible on this box (cards visible:',
          len(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')), ')')
samples, stop = [], False
TEMPS = sorted(k for k in F if k.startswith('temp_'))
def loop():
    pw = F.get('power1_average') or F.get('power1_input', '')
    while not stop:
        samples.append((time.time(), rd(pw), rd(F.get('freq1_input', '')), [rd(F[k]) for k in TEMPS],
                        rd(F.get('in0_input', '')), rd(F.get('freq2_input', ''))))
        time.sleep(0.004)
t = threading.Thread(target=loop); t.start()
args = sys.argv[1:] or ['--no-cpu-baseline', '--no-extras', '--steps', '60', '--warmup', '3']
r = subprocess.run([sys.executable, 'bench.py'] + args, capture_output=True, text=True)
stop = True; t.join()
print(r.stdout.strip().splitlines()[-1][:300] if r.stdout.strip() else r.stderr[-500:])
print('power cap (W):', (rd(F['power1_cap']) or 0) / 1e6 if 'power1_cap' in F else None)
# the timed steps are the last part of the run: take the last 30 % of the samples
n = len(samples); tail = samples[int(n * 0.7):]
pw = [s[1] / 1e6 for s in tail if s[1]]; fq = [s[2] / 1e6 for s in tail if s[2]]
if pw: print('power W (last 30 %% of the run = timed steps, %d samples): min %.0f median %.0f max %.0f' % (len(pw), min(pw), st.median(pw), max(pw)))
if fq: print('sclk MHz (same window): min %.0f median %.0f max %.0f' % (min(fq), st.median(fq), max(fq)))
for i, k in enumerate(TEMPS):
    tv = [s[3][i] / 1e3 for s in tail if s[3][i]]
    lims = {q.split('_')[-1]: (rd(F[q]) or 0) / 1e3 for q in F if q.startswith('limit_' + k[5:] + '_')}
    if tv: print('%s C (same window): min %.0f median %.0f max %.0f   limits %s' % (k[5:], min(tv), st.median(tv), max(tv), lims))
vv = [s[4] for s in tail if s[4]]
if vv: print('vddgfx mV (same window): min %d median %d max %d' % (min(vv), st.median(vv), max(vv)))
mc = [s[5] / 1e6 for s in tail if s[5]]
if mc: print('mclk MHz (same window): min %.0f median %.0f max %.0f' % (min(mc), st.median(mc), max(mc)))
pw_all = [s[1] / 1e6 for s in samples if s[1]]
if pw_all: print('power W over the whole run: min %.0f max %.0f' % (min(pw_all), max(pw_all)))
