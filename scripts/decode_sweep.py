import sys,re
for line in sys.stdin:
    if "|" not in line: print(line.rstrip()); continue
    head,_,rest=line.rpartition("|")
    items=re.findall(r"(\d+):(\d+)",rest)
    d={}
    for k,v in items:
        k=int(k); d[(k&255,k>>8)]=int(v)
    name=line[:28]
    cfgs=sorted({c for c,s in d if c})
    print(name)
    for c in cfgs:
        print("   cfg %2d: "%c+" ".join("x%d:%d"%(s,d[(c,s)]) for (cc,s) in sorted(d) if cc==c))
