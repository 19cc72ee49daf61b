import sys, collections
rows=[list(map(int,l.split())) for l in open(sys.argv[1])]
t0=min(r[1] for r in rows)
percu=collections.defaultdict(list)
for r in rows:
    b=r[0]; start=r[1]-t0; hw=r[2]; sec=r[3]
    xcc=hw>>32; cu=(hw>>8)&0xff
    tiles=[]
    for it in range(7):
        k0,k1,e1,cyc=r[5+it*4:5+it*4+4]
        if k0: tiles.append(((k0-t0)/100.0,(k1-t0)/100.0,(e1-t0)/100.0,cyc))
    percu[(xcc,cu)].append((b,sec,start/100.0,tiles))
import statistics
cnt=collections.Counter(len(v) for v in percu.values())
print("workgroups per CU histogram:",dict(cnt), "CUs:",len(percu))
ks=[];es=[];pro=[]
for key,v in percu.items():
    for b,sec,st,tiles in v:
        prev=None
        for (k0,k1,e1,_) in tiles:
            ks.append(k1-k0); es.append(e1-k1)
            if prev is not None: pro.append(k0-prev)
            prev=e1
print("K loop us: median %.2f min %.2f max %.2f | epilogue us: median %.2f min %.2f max %.2f | prologue us median %.2f"%(statistics.median(ks),min(ks),max(ks),statistics.median(es),min(es),max(es),statistics.median(pro)))
for key in list(percu)[:3]:
    print("CU",key)
    for b,sec,st,tiles in percu[key]:
        print("  block",b,"second",sec,"start %.2f"%st," ".join("[K %.1f-%.1f E-%.1f]"%t[:3] for t in tiles))
        print("     MHz between tile starts:"," ".join("%.0f"%((tiles[i+1][3]-tiles[i][3])/(tiles[i+1][0]-tiles[i][0])) for i in range(len(tiles)-1)))
end=max(t[2] for v in percu.values() for _,_,_,ts in v for t in ts)
print("last epilogue end us: %.1f"%end)
