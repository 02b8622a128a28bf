import sqlite3,sys,glob
for path in sys.argv[1:]:
    f=glob.glob(path+'/**/*.db',recursive=True)[0]
    db=sqlite3.connect(f); cur=db.cursor()
    print('==',path)
    rows=list(cur.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc limit 14"))
    for r in rows: print("  %-60s n=%5d avg %8.1f us total %8.2f ms"%(r[0][:60],r[1],r[2]/1e3,r[3]/1e6))
