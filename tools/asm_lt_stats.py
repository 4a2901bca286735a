"""Register, LDS and instruction statistics of k_vocoder_lt<35,3> from a `hipcc -save-temps` assembly file:
   python tools/asm_lt_stats.py FILE.s [kernel substring]
Prints VGPRs / scratch / LDS of the kernel and the instruction mix of its largest innermost loop (the two-sample body)."""
import re
import sys
from collections import Counter

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_vocoder_ltILi35ELi3"
s = open(path).read()
m = re.search(r"^(_Z\S*%s\S*):" % re.escape(pat), s, re.M)
name = m.group(1)
i = m.start()
j = s.index(".end_amdhsa_kernel", i)
body = s[i:j]
for key in ("next_free_vgpr", "accum_offset", "private_segment_fixed_size", "group_segment_fixed_size"):
    mm = re.search(r"\.amdhsa_%s (\d+)" % key, body)
    print(key, mm.group(1) if mm else None)
code = body[:body.index(".amdhsa_kernel")]
# split in basic blocks at labels; the loop body is the block(s) between a label and the branch back to it
lines = code.splitlines()
labels = {l.split(":")[0]: k for k, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
spans = []
for k, l in enumerate(lines):
    mm = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch (\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
        spans.append((labels[mm.group(1)], k))
inner = [a for a in spans if not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in spans)]
best = max(inner, key=lambda a: a[1] - a[0])
blk = lines[best[0]:best[1] + 1]
ins = [l.split()[0] for l in blk if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = Counter(ins)
valu = sum(v for k, v in c.items() if k.startswith("v_"))
lds = sum(v for k, v in c.items() if k.startswith("ds_"))
print("largest innermost loop: %d instructions, VALU %d, LDS %d, s_waitcnt %d, s_nop %d, vmem %d, scratch %d" % (
    len(ins), valu, lds, c["s_waitcnt"], c["s_nop"],
    sum(v for k, v in c.items() if k.startswith(("global_", "buffer_"))),
    sum(v for k, v in c.items() if k.startswith("scratch_"))))
print(dict(c.most_common(40)))
