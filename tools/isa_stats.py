"""Developer helper: instruction histogram and resource usage of one kernel in a hipcc -S listing.
    python tools/isa_stats.py <listing.s> <mangled-name-prefix> [--dump]"""
import collections, re, sys
t = open(sys.argv[1]).read().splitlines()
name = sys.argv[2]
s = next(i for i, l in enumerate(t) if l.startswith(name) and ":" in l)
e = next(i for i in range(s, len(t)) if "s_endpgm" in t[i])
body = t[s:e + 1]
ins = [l.split()[0] for l in body if l.startswith("\t") and l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".")]
h = collections.Counter(ins)
meta = [l.strip() for l in t[e:e + 120] if re.match(r"\s*; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|SGPRSpill|VGPRSpill)", l)]
print(name, len(ins), "instructions;", "; ".join(meta))
print("  ", h.most_common(40))
if "--dump" in sys.argv:
    print("\n".join(body))
