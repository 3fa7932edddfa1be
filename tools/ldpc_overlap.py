import sqlite3,glob,sys
db=glob.glob(sys.argv[1]+'/**/*.db',recursive=True)[0]
c=sqlite3.connect(db)
rows=c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
ld=[(s,e,st) for n,s,e,st in rows if 'ldpc_decode2' in n]
tot=sum(e-s for s,e,_ in ld); span=ld[-1][1]-ld[0][0]
ev=sorted([(s,1) for s,e,_ in ld]+[(e,-1) for s,e,_ in ld]); cur=0; busy=0; last=None
for t,d in ev:
    if cur>0: busy+=t-last
    cur+=d; last=t
print(len(ld),"ldpc launches; sum dur %.1f ms, union %.1f ms, span %.1f ms"%(tot/1e6,busy/1e6,span/1e6))
