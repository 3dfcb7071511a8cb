"""Compares two outputs of isa_stats.py kernel by kernel (names normalised for StaticCfg's trailing PIPE_ argument, which round 5 added):
    python profiles/tools/isa_cmp.py profiles/r05_isa_stats_r04_tree.txt profiles/r05_isa_stats_r05_tree.txt
prints every kernel whose instruction count / registers / scratch / occupancy / opcode-sequence hash differ, and how many are identical."""
import re,sys
def load(f):
    d={}
    for l in open(f):
        name,rest=l.rsplit(": inst",1)
        name=re.sub(r"(StaticCfg<(?:[^<>]*?))(, 0)>", lambda m: m.group(1)+">" if m.group(1).count(",")==11 else m.group(0), name)
        d[name]=rest.strip()
    return d
a,b=load(sys.argv[1]),load(sys.argv[2])
same=0
for k in a:
    if k not in b: print("MISSING",k); continue
    if a[k]==b[k]: same+=1
    else: print(k,"\n   base:",a[k],"\n   new: ",b[k])
print(same,"identical of",len(a))
